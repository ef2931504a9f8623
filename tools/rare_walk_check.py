import os, sys, ctypes as C
sys.path.insert(0, os.getcwd())
from charls_amd import capi, synth
lib = capi.load_product()
raw = lib.lib
raw.charls_amd_speculation_counters.argtypes = [C.POINTER(C.c_uint64), C.c_int32]
img = synth.frame_numpy(4096, 4096, seed=2, bits=8)
def enc(tag):
    before = (C.c_uint64 * 6)(); raw.charls_amd_speculation_counters(before, 6)
    b = bytes(lib.encode(img, width=4096, height=4096, bits_per_sample=8))
    after = (C.c_uint64 * 6)(); raw.charls_amd_speculation_counters(after, 6)
    print(tag, len(b), [int(after[i] - before[i]) for i in range(6)], flush=True)
    return b
a = enc("default")
capi.set_knob("RARE_WARM_EVENTS", 0)
b = enc("rare warm 0")
capi.set_knob("RARE_WARM_EVENTS", 100000000)
c = enc("rare warm all")
print("equal:", a == b == c)
