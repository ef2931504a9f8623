"""Decode throughput of the speed-path kernels over batch sizes (batch API, frames resident in HBM).
Run on the GPU box: python tools/decode_sweep.py [--frames 4096] [--groups 0,16] [--sizes 64,256,1024,4096] [--bits 8]
CHARLS_AMD_DECODE_GROUP: 0 = one scan per wavefront (scan_fast_decode.hip), 8/16/32 = lanes per scan (scan_group_decode.hip)."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from charls_amd import batch, capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--width", type=int, default=4096)
ap.add_argument("--height", type=int, default=4096)
ap.add_argument("--bits", type=int, default=8)
ap.add_argument("--groups", default="0,16", help="lanes per scan, optionally with the wavefronts per workgroup: 8,16:4,32:8")
ap.add_argument("--sizes", default="64,256,1024,4096")
ap.add_argument("--repeat", type=int, default=1)
ap.add_argument("--distinct", type=int, default=0,
                help="synthesise only this many distinct frames and repeat them (few torch kernels: rocprofv3 --pmc crashes in "
                     "torch's synthesis kernels when thousands of them run under it)")
ap.add_argument("--encode-repeat", type=int, default=0, help="time this many further encodes of the batch (best is printed)")
ap.add_argument("--lib", default=None, help="another build of the product library (A/B runs of kernel variants)")
ap.add_argument("--kind", default="gradient", help="gradient (the bench's frames), mixed, hard, zero, or tulips (the reference's natural image tiled, bench.py: data_frames)")
args = ap.parse_args()
if args.lib:
    capi.PRODUCT_LIB = os.path.abspath(args.lib)

import glob
import threading


class ClockSampler:
    """Samples the shader clock (sysfs pp_dpm_sclk, the line marked '*') while a decode runs."""

    def __init__(self):
        self.files = glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk")
        self.values = []
        self.stop = False

    def _run(self):
        while not self.stop:
            for f in self.files:
                try:
                    for line in open(f):
                        if "*" in line:
                            self.values.append(int(line.split(":")[1].strip().lower().split("mhz")[0]))
                except Exception:
                    pass
            time.sleep(0.05)

    def __enter__(self):
        self.thread = threading.Thread(target=self._run, daemon=True)
        self.thread.start()
        return self

    def __exit__(self, *a):
        self.stop = True
        self.thread.join()

    def mean(self):
        return round(sum(self.values) / len(self.values)) if self.values else None


lib = capi.load_product()
dev = torch.device("cuda:0")
t0 = time.perf_counter()
if args.kind == "tulips":
    import bench
    bench.WIDTH, bench.HEIGHT = args.width, args.height
    frames = torch.empty((args.frames, args.height, args.width), dtype=torch.uint8, device=dev)
    bench.data_frames(torch, "tulips", args.frames, dev, frames)
elif args.distinct and args.distinct < args.frames:
    base = synth.frames_torch(args.distinct, args.width, args.height, seed0=2, bits=args.bits, kind=args.kind, device=dev)
    frames = base.repeat((args.frames + args.distinct - 1) // args.distinct, 1, 1)[:args.frames].contiguous()
    del base
else:
    frames = synth.frames_torch(args.frames, args.width, args.height, seed0=2, bits=args.bits, kind=args.kind, device=dev)
torch.cuda.synchronize()
out = torch.empty_like(frames)
batch.set_workspace_limit(64 << 30, lib)
t1 = time.perf_counter()
enc = batch.encode_batch(frames, bits_per_sample=args.bits, lib=lib)
torch.cuda.synchronize()
t2 = time.perf_counter()
assert (enc.errcs == 0).all()
if args.encode_repeat:
    best = None
    for _ in range(args.encode_repeat):
        torch.cuda.synchronize()
        a = time.perf_counter()
        again = batch.encode_batch(frames, bits_per_sample=args.bits, lib=lib)
        torch.cuda.synchronize()
        b = time.perf_counter()
        best = b - a if best is None else min(best, b - a)
        assert (again.errcs == 0).all() and (again.sizes == enc.sizes).all()
        del again
    print(f"encode again: best of {args.encode_repeat} {best:.3f}s = {args.width * args.height / 1e6 * args.frames / best:.0f} MPix/s", flush=True)
batch.release_work_areas(lib)
mpix = args.width * args.height / 1e6
print(f"synth {t1 - t0:.1f}s encode {t2 - t1:.2f}s = {mpix * args.frames / (t2 - t1):.0f} MPix/s", flush=True)
rows = []
for spec in args.groups.split(","):
    g, wg = (int(x) for x in (spec.split(":") + ["1"])[:2])
    if g < 0:  # the library's own rule
        g = wg = None
    capi.set_knob("DECODE_GROUP", g)
    capi.set_knob("DECODE_WORKGROUP_WAVES", wg)
    for n in [int(x) for x in args.sizes.split(",")]:
        if n > args.frames:
            continue
        best = None
        for it in range(args.repeat + 1):  # first pass warms allocations
            out[:n].zero_()
            torch.cuda.synchronize()
            a = time.perf_counter()
            with ClockSampler() as clocks:
                params, errcs, gpu_ms = batch.decode_batch(enc.streams[:n], enc.sizes[:n], out[:n], lib=lib)
                torch.cuda.synchronize()
            b = time.perf_counter()
            best = b - a if best is None or (it > 0 and b - a < best) or it == 1 else best
        ok = bool((errcs == 0).all())
        for f0 in range(0, n, 64):  # compare in pieces: torch.equal materialises a mask of the operands' size
            ok = ok and torch.equal(out[f0:min(n, f0 + 64)], frames[f0:min(n, f0 + 64)])
        rows.append({"group": g, "workgroup_waves": wg, "frames": n, "decode_s": round(best, 4), "mpix_s": round(mpix * n / best, 1),
                     "kernel_ms": round(gpu_ms[0], 2), "sclk_mhz": clocks.mean(), "ok": ok})
        print(json.dumps(rows[-1]), flush=True)
