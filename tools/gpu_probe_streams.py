"""Scratch: decode scaling with the number of concurrent streams (2048x2048 frames)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from charls_amd import batch, capi, synth
lib = capi.load_product()
w = 2048
for n in [int(a) for a in sys.argv[1:]] or [64, 256, 1024, 2048]:
    frames = synth.frames_torch(min(n, 64), w, w, seed0=2, device="cuda:0")
    if n > 64:
        frames = frames.repeat((n + 63) // 64, 1, 1)[:n].contiguous()
    torch.cuda.synchronize()
    t0 = time.perf_counter(); enc = batch.encode_batch(frames); t1 = time.perf_counter()
    out = torch.empty_like(frames)
    t1 = time.perf_counter(); _, errcs, dt = batch.decode_batch(enc.streams, enc.sizes, out); t2 = time.perf_counter()
    ok = bool((enc.errcs == 0).all() and (errcs == 0).all() and torch.equal(out, frames))
    mp = n * w * w / 1e6
    print(f"streams={n}: enc {1e3*(t1-t0):.0f} ms stages={[round(v,1) for v in enc.gpu_ms[2:]]}  dec {1e3*(t2-t1):.0f} ms ({mp/(t2-t1):.0f} MPix/s) ok={ok}", flush=True)
    del frames, out, enc
    torch.cuda.empty_cache()
