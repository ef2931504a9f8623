"""How often a wavefront of scan_group_decode.hip takes each of its paths, on the bench's synthetic frames.
CPU only: the kernel source runs in the thread-per-lane harness of tests/emu with its path counters on; the counts are
then priced with the instruction counts of the compiled gfx950 code.
    python tools/decode_path_profile.py [--width 4096] [--lines 24] [--group 8]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import emu_bind  # noqa: E402
import jls_container  # noqa: E402
import oracle_bind as ob  # noqa: E402
from charls_amd import synth  # noqa: E402
from test_emu_serial_kernels import _stream_copy  # noqa: E402

NAMES = ["rounds", "refills", "line starts", "step loops", "trips of the step-count loop", "steps", "general run handler",
         "run-length code bits", "trips of the run fill", "unusual codes", "line ends", "refills byte by byte",
         "trips of the refill's delete loop", "(retired: handler of empty runs)", "calls of prepare (64 entries of the previous line)",
         "run handler out of one 64-bit window", "run services inside the step loop"]

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=4096)
ap.add_argument("--lines", type=int, default=24)
ap.add_argument("--group", type=int, default=8)
ap.add_argument("--kind", default="gradient")
ap.add_argument("--near", type=int, default=0, help="NEAR of the scans (decode_scans_group<.., kNear = true>)")
args = ap.parse_args()

L = emu_bind.profile_lib()
count = 64 // args.group  # one wavefront
keep, outs, descs, imgs = [], [], [], []
for f in range(count):
    if args.kind == "tulips":  # the reference's natural test image, tiled to the width, a different band of lines per scan
        import common
        tile, _ = common.read_pnm("tulips-gray-8bit-512-512.pgm")
        band = np.roll(tile, shift=(-(37 * f) % 512, 29 * f), axis=(0, 1))[:args.lines]
        img = np.ascontiguousarray(np.tile(band, (1, (args.width + 511) // 512))[:, :args.width])
    else:
        img = synth.frame_numpy(args.width, args.lines, seed=1000 + f, bits=8, kind=args.kind)
    jls = ob.encode(img, width=args.width, height=args.lines, bits_per_sample=8, near_lossless=args.near)
    cont = jls_container.parse(jls)
    pc = jls_container.validated_pc(cont.pc, cont.bits, args.near)
    if args.near:
        img = ob.decode(jls)[1].reshape(img.shape)  # what a decoder has to produce
    pix = np.zeros(args.width * args.lines, dtype=np.uint8)
    descs.append(emu_bind.make_desc(args.width, args.lines, 1, 0, 8, args.near, 0, pc, 0, pix, args.width,
                                    _stream_copy(jls, cont.scans[0].data_start), keep))
    outs.append(pix)
    imgs.append(img)
arr = (emu_bind.ScanDesc * count)(*descs)
res = (emu_bind.ScanResult * count)()
counts = (C.c_ulonglong * 32)()
assert L.emu_profile_decode_group(arr, res, count, args.group, counts) == 0
for r, o, img in zip(res, outs, imgs):
    assert (r.errc, r.flags) == (0, 0)
    assert o.tobytes() == np.ascontiguousarray(img).tobytes()
samples = args.width * args.lines
print(f"one wavefront, {count} scans of {args.width} x {args.lines} ({samples} samples each), {args.group} lanes per scan, kind {args.kind}")
for i, n in enumerate(NAMES):
    print(f"  {n:32s} {counts[i]:10d}   per 1000 samples of a scan: {1000.0 * counts[i] / samples:8.2f}")
