"""Where the time of the pipelined host-buffer batch (bench.py: host_abi) goes.  Run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from charls_amd import batch, capi, synth
lib = capi.load_product()
dev = torch.device("cuda:0")
n, chunk, W = 256, 64, 4096
batch.set_workspace_limit(96 << 30, lib)
frames = synth.frames_torch(n, W, W, seed0=2, bits=8, device=dev)
pitch = (batch.estimated_destination_size(W, W, 8, 1) + 255) & ~255
streams = torch.empty((n, pitch), dtype=torch.uint8, device=dev)
out = torch.empty_like(frames)
host_frames = torch.empty((n, W, W), dtype=torch.uint8).pin_memory(); host_frames.copy_(frames)
host_streams = torch.empty((n, pitch), dtype=torch.uint8).pin_memory()
def t(f, name):
    torch.cuda.synchronize(); a = time.perf_counter(); r = f(); torch.cuda.synchronize(); print(f"{name}: {(time.perf_counter()-a)*1e3:.1f} ms", flush=True); return r
batch.encode_batch(out[:chunk], streams=streams[:chunk], lib=lib)
t(lambda: out.copy_(host_frames, non_blocking=True), "H2D 256 frames (4.3 GB) contiguous")
t(lambda: out[:chunk].copy_(host_frames[:chunk], non_blocking=True), "H2D 64 frames")
e = t(lambda: batch.encode_batch(out[:chunk], streams=streams[:chunk], lib=lib), "encode 64 frames")
e = t(lambda: batch.encode_batch(out[:chunk], streams=streams[:chunk], lib=lib), "encode 64 frames again")
w = int(e.sizes.max())
t(lambda: host_streams[:chunk, :w].copy_(streams[:chunk, :w], non_blocking=True), "D2H 64 streams, strided 2-D copy")
def rows():
    for f in range(chunk):
        host_streams[f, :int(e.sizes[f])].copy_(streams[f, :int(e.sizes[f])], non_blocking=True)
t(rows, "D2H 64 streams, one copy per stream")
e2 = t(lambda: batch.encode_batch(out, streams=streams, lib=lib), "encode 256 frames in one call")
e2 = t(lambda: batch.encode_batch(out, streams=streams, lib=lib), "encode 256 frames in one call again")
