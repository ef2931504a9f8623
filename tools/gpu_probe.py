"""Scratch GPU probe used during development: times the C-ABI and batch paths at a few sizes."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from charls_amd import batch, capi, synth
lib = capi.load_product()
print("device", torch.cuda.get_device_name(0), "status", lib.lib.charls_amd_device_status(), flush=True)
cfgs = [(1, 1024, 8), (64, 1024, 8), (1, 4096, 8), (16, 4096, 8), (64, 4096, 8), (8, 4096, 16)]
if len(sys.argv) > 1:
    cfgs = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]]
for n, w, bits in cfgs:
    frames = synth.frames_torch(n, w, w, seed0=2, bits=bits, device="cuda:0")
    torch.cuda.synchronize()
    for engine in (2, 1) if n * w * w <= 64 * 1024 * 1024 else (2,):
        batch.set_encode_engine(engine)
        for rep in range(2):
            t0 = time.perf_counter(); enc = batch.encode_batch(frames, bits_per_sample=bits); t1 = time.perf_counter()
        mp = n * w * w / 1e6
        print(f"frames={n} {w}x{w}x{bits} engine={engine}: enc {1e3*(t1-t0):.1f} ms ({mp/(t1-t0):.1f} MPix/s) stages={[round(v,2) for v in enc.gpu_ms]}", flush=True)
    batch.set_encode_engine(0)
    out = torch.empty_like(frames)
    for rep in range(2):
        t1 = time.perf_counter(); _, errcs, dt = batch.decode_batch(enc.streams, enc.sizes, out); t2 = time.perf_counter()
    ok = bool((enc.errcs == 0).all() and (errcs == 0).all() and torch.equal(out, frames))
    print(f"frames={n} {w}x{w}x{bits}: dec {1e3*(t2-t1):.1f} ms ({mp/(t2-t1):.1f} MPix/s, kernels {[round(v,2) for v in dt]}) ok={ok} bytes/frame={int(enc.sizes.mean())}", flush=True)
