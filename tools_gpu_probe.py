"""Scratch GPU probe used during development: times the C-ABI and batch paths at a few sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
import numpy as np, torch
from charls_amd import batch, capi, synth
lib = capi.load_product()
print("device", torch.cuda.get_device_name(0), "status", lib.lib.charls_amd_device_status(), flush=True)
for n, w in [(1, 1024), (16, 1024), (64, 1024), (256, 1024), (8, 4096)]:
    frames = synth.frames_torch(n, w, w, seed0=2, device="cuda:0")
    torch.cuda.synchronize()
    t0 = time.perf_counter(); enc = batch.encode_batch(frames); t1 = time.perf_counter()
    out = torch.empty_like(frames)
    _, errcs, dt = batch.decode_batch(enc.streams, enc.sizes, out); t2 = time.perf_counter()
    ok = bool((enc.errcs == 0).all() and (errcs == 0).all() and torch.equal(out, frames))
    mp = n * w * w / 1e6
    print(f"frames={n} {w}x{w}: enc {1e3*(t1-t0):.1f} ms ({mp/(t1-t0):.1f} MPix/s, kernels {enc.gpu_ms}) "
          f"dec {1e3*(t2-t1):.1f} ms ({mp/(t2-t1):.1f} MPix/s, kernels {dt}) ok={ok}", flush=True)
