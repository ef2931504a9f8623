"""ONE 4096 x 4096 8-bit frame through the host-pointer C ABI (PCIe inclusive): encode latency, best and median of N calls.
Run on the GPU box: python tools/one_frame_latency.py [--calls 12] [--no-decode]"""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from charls_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--calls", type=int, default=12)
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--decode", action="store_true")
ap.add_argument("--lib", default=None, help="another build of the product library (A/B runs)")
args = ap.parse_args()
if args.lib:
    capi.PRODUCT_LIB = os.path.abspath(args.lib)
lib = capi.load_product()
img = synth.frame_numpy(args.size, args.size, seed=2, bits=8)
import numpy as np  # noqa: E402
dst = np.empty(args.size * args.size + args.size * args.size // 16 + 2048, dtype=np.uint8)  # allocated outside the clock (cli/benchmark.cpp:60-90)
lib.encode(img, width=args.size, height=args.size, bits_per_sample=8, destination=dst)  # warm-up (allocations, module load)
times = []
for _ in range(args.calls):
    a = time.perf_counter()
    jls = lib.encode(img, width=args.size, height=args.size, bits_per_sample=8, destination=dst)
    times.append((time.perf_counter() - a) * 1e3)
jls = jls.tobytes()
print(f"encode_from_buffer, {args.size} x {args.size}: best {min(times):.2f} ms, median {statistics.median(times):.2f} ms, {len(jls)} bytes", flush=True)
if args.decode:
    a = time.perf_counter()
    _, px = lib.decode(jls)
    print(f"decode_to_buffer: {(time.perf_counter() - a) * 1e3:.1f} ms, pixels {'equal' if px.tobytes() == img.tobytes() else 'DIFFER'}")
