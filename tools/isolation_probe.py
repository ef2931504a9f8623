"""While ONE 4096 x 4096 stream decodes (a kernel that runs for 3.4 s), what does another thread's encode call wait for?
CHARLS_AMD_TRACE=1 python tools/isolation_probe.py   (stderr: where every call's time went)"""
import ctypes as C
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from charls_amd import capi, synth  # noqa: E402

lib = capi.load_product()
hip = C.CDLL("libamdhip64.so")
lo, hi = C.c_int(0), C.c_int(0)
print("hipDeviceGetStreamPriorityRange:", hip.hipDeviceGetStreamPriorityRange(C.byref(lo), C.byref(hi)), "lowest", lo.value, "highest", hi.value, flush=True)
print("GPU_MAX_HW_QUEUES =", os.environ.get("GPU_MAX_HW_QUEUES"), flush=True)
big = synth.frame_numpy(4096, 4096, seed=2)
big_jls = lib.encode(big, width=4096, height=4096)
lib.decode(lib.encode(synth.frame_numpy(256, 256, seed=1), width=256, height=256))
for trial in range(2):
    t = threading.Thread(target=lambda: lib.decode(big_jls))
    t0 = time.perf_counter()
    t.start()
    time.sleep(0.5)
    lat = []
    for _ in range(3):
        a = time.perf_counter()
        lib.encode(big, width=4096, height=4096)
        lat.append((time.perf_counter() - a) * 1e3)
    t.join()
    print(f"trial {trial}: decode {time.perf_counter() - t0:.2f} s; encodes beside it: {', '.join(f'{v:.1f}' for v in lat)} ms", flush=True)
