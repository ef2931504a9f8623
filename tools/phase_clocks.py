#!/usr/bin/env python3
"""Where the wavefronts of sort_tiles / pack_tiles spend their clocks (a debug build: the marks of JLS_PHASE in
tile_pipeline.hip count only when the library is built with CHARLS_AMD_CXXFLAGS=-DJLS_PHASE_CLOCKS).

    CHARLS_AMD_CXXFLAGS=-DJLS_PHASE_CLOCKS python charls_amd/build.py --force      (here)
    gpurun -- python tools/phase_clocks.py                                          (on the GPU box)

Runs one encode step of bench.py's workload with 512 frames, sums the `phase_clocks` lines the library prints per call
and prints each phase's share of its kernel's wavefront clocks (of a sample: every 64th workgroup counts -- with all of
them counting, 75 M atomics on 22 addresses queue up in front of the kernels' own memory traffic and the shares
measure the atomics).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
NAMES = {
    "sort_tiles": {0: "lines requested", 1: "tables zeroed + barrier (lines arrive)", 2: "P1 keys, histogram", 3: "barrier",
                   4: "offsets", 5: "barrier", 6: "P2 ranks, records", 7: "barrier", 8: "P3 pieces out"},
    "pack_tiles": {9: "slot map requested, piece table", 10: "barrier", 11: "row table + barrier", 12: "code words into LDS",
                   13: "barrier", 14: "bits of the thread's samples", 15: "scan of the bit counts", 16: "look-back (wavefront 0)",
                   17: "barrier (the others wait for the look-back)", 18: "bits into LDS", 19: "barrier", 20: "barrier + tails (thread 0)", 21: "words out"},
}


def main() -> None:
    frames = sys.argv[1] if len(sys.argv) > 1 else "512"
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--frames", frames, "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
           "--no-extras"]
    run = subprocess.run(cmd, capture_output=True, text=True)
    total = [0] * 28
    calls = 0
    for line in run.stderr.splitlines():
        if line.startswith("phase_clocks"):
            calls += 1
            for i, v in enumerate(line.split()[1:]):
                total[i] += int(v)
    if calls == 0:
        sys.exit("no phase_clocks lines: the library is not a -DJLS_PHASE_CLOCKS build\n" + run.stderr[-2000:])
    print(f"{calls} calls of the tile pipeline")
    for kernel, names in NAMES.items():
        whole = sum(total[i] for i in names) or 1
        print(f"{kernel}: {whole / 1e9:.1f} G clocks of wavefront time")
        for i, name in names.items():
            print(f"  {100.0 * total[i] / whole:5.1f} %  {name}")


if __name__ == "__main__":
    main()
