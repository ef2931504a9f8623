"""`threads_abi` of bench.py on its own (no 210 GB of bench frames around it): T host threads x one handle per call through the
48-symbol ABI, host buffers in and out, the harness of cpu_baseline.all_cores.

    python tools/threads_abi_probe.py [--threads 256,512] [--seconds 12]"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch  # noqa: E402

import bench  # noqa: E402
from charls_amd import batch, capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", default="256,512")
ap.add_argument("--seconds", type=float, default=12.0)
ap.add_argument("--distinct", type=int, default=256)
ap.add_argument("--stagger", type=float, default=0.0, help="threads start up to this many seconds apart and idle a random while between calls")
ap.add_argument("--workspace-gib", type=int, default=96)
args = ap.parse_args()

lib = capi.load_product()
batch.set_workspace_limit(args.workspace_gib << 30, lib)
frames = synth.frames_torch(args.distinct, bench.WIDTH, bench.HEIGHT, seed0=2, bits=8, device="cuda:0")
host = frames.cpu().numpy()
del frames
torch.cuda.empty_cache()
host_list = [host[i] for i in range(args.distinct)]
pitch = (batch.estimated_destination_size(bench.WIDTH, bench.HEIGHT, 8, 1) + 255) & ~255
for t in [int(x) for x in args.threads.split(",")]:
    row = bench.threads_abi(lib, host_list, pitch, t, args.seconds, stagger=args.stagger)
    if args.stagger:
        row["stagger_s"] = args.stagger
    print(json.dumps(row), flush=True)
    print(f"# work areas held after the run: {batch.work_area_bytes(lib) / 2**30:.1f} GiB", flush=True)
batch.release_work_areas(lib)
