#!/usr/bin/env python3
"""Turns gpurun_out/final/ (written by tools/collect_profiles.sh on the GPU box) into the files kept under profiles/."""
import csv
import json
import os
import re
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", sys.argv[2] if len(sys.argv) > 2 else "final")
DST = os.path.join(ROOT, "profiles")
TAG = sys.argv[1] if len(sys.argv) > 1 else "r01"
PMC_FRAMES = 64


def short(name):
    m = re.search(r"(jls::[\w:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def counters(sub):
    """kernel -> counter -> (sum, launches)"""
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    with open(os.path.join(SRC, sub, "p_counter_collection.csv"), newline="") as f:
        for row in csv.DictReader(f):
            k = short(row["Kernel_Name"])
            if not k.startswith("jls::"):
                continue
            acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
            launches[k].add(row["Dispatch_Id"])
    return acc, {k: len(v) for k, v in launches.items()}


def main():
    with open(os.path.join(SRC, "bench.json")) as f:
        bench = json.loads(f.read().strip().splitlines()[-1])
    with open(os.path.join(DST, f"{TAG}_bench_default_1gpu.json"), "w") as f:
        json.dump(bench, f, indent=1)
        f.write("\n")

    # ---- kernel statistics of `python bench.py --no-cpu-baseline` (same workload as the bench line)
    rows = []
    with open(os.path.join(SRC, "stats", "bench_kernel_stats.csv"), newline="") as f:
        for row in csv.DictReader(f):
            rows.append(row)
    with open(os.path.join(DST, f"{TAG}_rocprof_kernel_stats_bench_default.csv"), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --no-cpu-baseline "
                f"(frames_per_gpu={bench['config']['frames_per_gpu']}, warmup 1 + 2 steps; torch kernels = synthetic frame generation)\n")
        f.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,percent\n")
        for r in rows[:24]:
            f.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e6:.3f},"
                    f"{float(r['MinNs'])/1e6:.3f},{float(r['MaxNs'])/1e6:.3f},{r['Percentage']}\n")

    # ---- HBM traffic per kernel (separate PMC passes)
    fetch, lf = counters("pmc_FETCH_SIZE")
    write, _ = counters("pmc_WRITE_SIZE")
    per_kernel = {}
    with open(os.path.join(DST, f"{TAG}_pmc_traffic_frames{PMC_FRAMES}.csv"), "w") as f:
        f.write(f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes): python bench.py --frames {PMC_FRAMES} --steps 1 --warmup 0 --no-cpu-baseline\n")
        f.write("# units: counters are in KiB (x1024 = bytes); FETCH_SIZE is doubled as /opt/skills/guides/MI355X_MICROARCH.md (HBM) prescribes for gfx950\n")
        f.write("kernel,launches,FETCH_SIZE_raw,WRITE_SIZE_raw,hbm_bytes_per_frame_corrected\n")
        for k in sorted(fetch, key=lambda k: -(2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"])):
            b = (2 * fetch[k]["FETCH_SIZE"] + write[k]["WRITE_SIZE"]) * 1024 / PMC_FRAMES
            per_kernel[k] = b
            f.write(f"{k},{lf[k]},{fetch[k]['FETCH_SIZE']:.1f},{write[k]['WRITE_SIZE']:.1f},{b:.0f}\n")
    dom = bench["roofline"]["kernel"]
    match = [k for k in per_kernel if dom.split("+")[0] in k]
    with open(os.path.join(DST, f"{TAG}_traffic.json"), "w") as f:
        json.dump({"kernel": dom, "hbm_bytes_per_frame": per_kernel[match[0]] if match else None,
                   "source": f"profiles/{TAG}_pmc_traffic_frames{PMC_FRAMES}.csv",
                   "note": f"(2*FETCH_SIZE + WRITE_SIZE) * 1024 / {PMC_FRAMES} frames; FETCH_SIZE doubled per the guide's gfx950 correction"},
                  f, indent=1)
        f.write("\n")

    # ---- instruction mix of the decoder
    inst, li = counters("pmc_inst")
    with open(os.path.join(DST, f"{TAG}_pmc_instructions_frames{PMC_FRAMES}.txt"), "w") as f:
        f.write(f"# rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE\n")
        f.write(f"# python bench.py --frames {PMC_FRAMES} --steps 1 --warmup 0 --no-cpu-baseline; sums over all launches of a kernel; "
                "per-sample figures divide by 64 frames x 4096 x 4096 samples (wave-level instruction counts; a wavefront of decode_scans_group carries several scans, so its per-sample figure is per wavefront-step divided by the scans it advances)\n")
        samples = PMC_FRAMES * 4096 * 4096
        for k in sorted(inst):
            c = inst[k]
            f.write(f"{k} launches={li[k]} " + " ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())) + "\n")
            if c.get("SQ_INSTS_VALU") and ("decode_scans" in k or "bias_chains" in k or "code_events" in k):
                f.write(f"    per sample: VALU {c['SQ_INSTS_VALU']/samples:.1f}  SALU {c['SQ_INSTS_SALU']/samples:.1f}  "
                        f"LDS {c['SQ_INSTS_LDS']/samples:.2f}  wave-cycles(x4) {4*c['SQ_WAVE_CYCLES']/samples:.0f}\n")
    print("profiles written for", TAG, "value", bench["value"])


if __name__ == "__main__":
    main()
