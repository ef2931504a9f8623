"""Scratch: one encode + one decode of a small batch, for rocprofv3 counter collection."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from charls_amd import batch, capi, synth
n, w = int(sys.argv[1]), int(sys.argv[2])
lib = capi.load_product()
frames = synth.frames_torch(n, w, w, seed0=2, device="cuda:0")
torch.cuda.synchronize()
enc = batch.encode_batch(frames)
out = torch.empty_like(frames)
t0 = time.perf_counter()
_, errcs, dt = batch.decode_batch(enc.streams, enc.sizes, out)
t1 = time.perf_counter()
print("decode ms", 1e3 * (t1 - t0), "ok", bool((errcs == 0).all() and torch.equal(out, frames)), flush=True)
