"""Soak test of the host-pointer ABI under threads: T threads code frames of a dozen geometries and modes in random order for a
while, every result compared with the oracle's (computed before the threads start).  Exercises the coalescer's keys, the pool of
handle resources (sets reused for other sizes: buffers grow while other batches run) and the shared work areas.

    python tools/threads_soak.py [--threads 64] [--seconds 60]"""
import argparse
import os
import random
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import oracle_bind as ob  # noqa: E402
from charls_amd import batch, capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--threads", type=int, default=64)
ap.add_argument("--seconds", type=float, default=60.0)
args = ap.parse_args()
lib = capi.load_product()

shapes = [(64, 64, 8, 1, 0, 0, 0), (300, 200, 8, 1, 0, 0, 0), (1024, 768, 8, 1, 0, 0, 0), (2048, 512, 8, 1, 0, 0, 0), (4096, 64, 8, 1, 0, 0, 0),
          (200, 144, 16, 1, 0, 0, 0), (640, 480, 12, 1, 0, 0, 0), (160, 120, 8, 3, 2, 0, 1), (256, 128, 8, 3, 1, 0, 0), (128, 96, 8, 3, 0, 0, 0),
          (256, 128, 8, 1, 0, 2, 0), (200, 100, 8, 3, 2, 3, 0), (9000, 3, 8, 1, 0, 0, 0), (333, 77, 8, 3, 1, 2, 0), (640, 100, 12, 1, 0, 3, 0),
          (512, 64, 8, 4, 1, 1, 0)]
cases = []
for i, (w, h, bits, comps, ilv, near, ct) in enumerate(shapes):
    for v in range(2):
        img = synth.frame_numpy(w, h, seed=1000 + 10 * i + v, bits=bits, components=comps, kind="mixed", interleaved=(ilv != 0))
        kw = dict(width=w, height=h, bits_per_sample=bits, component_count=comps, interleave_mode=ilv, near_lossless=near, color_transformation=ct)
        want = ob.encode(img, **kw)
        pixels = ob.decode(want)[1].tobytes() if near else np.ascontiguousarray(img).tobytes()
        cases.append((img, kw, want, pixels))
print(f"{len(cases)} cases, {args.threads} threads, {args.seconds:.0f} s", flush=True)
failures, done = [], [0] * args.threads
deadline = time.perf_counter() + args.seconds


def worker(t):
    rng = random.Random(t)
    try:
        while time.perf_counter() < deadline and not failures:
            img, kw, want, pixels = cases[rng.randrange(len(cases))]
            if rng.random() < 0.5:
                if lib.encode(img, **kw) != want:
                    failures.append(("encode", kw))
            else:
                if lib.decode(want)[1].tobytes() != pixels:
                    failures.append(("decode", kw))
            done[t] += 1
            if rng.random() < 0.02:
                batch.release_work_areas(lib)  # (another thread's release while batches run)
    except BaseException as e:  # noqa: BLE001
        failures.append(("exception", repr(e)))


pool = [threading.Thread(target=worker, args=(t,)) for t in range(args.threads)]
for th in pool:
    th.start()
for th in pool:
    th.join()
print("calls:", sum(done), "failures:", failures[:5], "engine counters:", capi.engine_counters(lib), flush=True)
sys.exit(1 if failures else 0)
