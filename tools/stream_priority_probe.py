"""Does the priority of the stream a decoder launch runs on change how long the kernel takes?  One / 64 frames of the bench's data
through the batch API on a torch stream, on streams made by hipStreamCreateWithFlags and by hipStreamCreateWithPriority(-1, 0, 1).
    python tools/stream_priority_probe.py"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from charls_amd import batch, capi, synth  # noqa: E402

hip = C.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
lib = capi.load_product()
frames = synth.frames_torch(64, 4096, 4096, seed0=2, bits=8, device="cuda:0")
enc = batch.encode_batch(frames, lib=lib)
out = torch.empty_like(frames)
least, greatest = C.c_int(), C.c_int()
hip.hipDeviceGetStreamPriorityRange(C.byref(least), C.byref(greatest))
print(f"priority range: least {least.value}, greatest {greatest.value}", flush=True)


def made(priority):
    s = C.c_void_p()
    rc = hip.hipStreamCreateWithFlags(C.byref(s), 1) if priority is None else hip.hipStreamCreateWithPriority(C.byref(s), 1, priority)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)


for name, stream in [("torch's current stream", torch.cuda.current_stream())] + [(f"hipStreamCreateWith{'Flags' if p is None else f'Priority({p})'}", made(p))
                                                                                    for p in (None, -1, 0, 1)]:
    for n in (1, 64):
        with torch.cuda.stream(stream):
            torch.cuda.synchronize()
            a = time.perf_counter()
            _, errcs, ms = batch.decode_batch(enc.streams[:n], enc.sizes[:n], out[:n], lib=lib)
            torch.cuda.synchronize()
            b = time.perf_counter()
        assert (errcs == 0).all()
        print(f"{name:36s} {n:3d} frames: {1e3 * (b - a):8.1f} ms", flush=True)
