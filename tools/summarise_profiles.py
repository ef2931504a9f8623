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
# the instantiation of the headline decoder that the bench's 4096-frame launch uses (runtime.hip: decode_group_plan)
DOMINANT = os.environ.get("CHARLS_AMD_DOMINANT_INSTANTIATION", "<unsigned char, 16, 1, 4, false>")
DOMINANT_KNOBS = "CHARLS_AMD_DECODE_GROUP=16 CHARLS_AMD_DECODE_WORKGROUP_WAVES=4"


def short(name):
    m = re.search(r"(jls::[\w:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def counters(sub):
    """kernel -> counter -> (sum, launches)"""
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    path = os.path.join(SRC, sub, "p_counter_collection.csv")
    if not os.path.exists(path):
        return acc, {}
    with open(path, newline="") as f:
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
        encoder = sum(b for k, b in per_kernel.items() if "tile::" in k or "stuff" in k or "place_" in k)
        f.write(f"# encoder (tile pipeline + stuffing + container kernels), sum: {encoder:.0f} bytes per frame\n")
    # ---- the dominant kernel with the bench's own number of frames: traffic and instruction counts per launch
    dom = bench["roofline"]["kernel"]
    frames = bench["config"]["frames_per_gpu"]
    fetch_big, lb = counters("pmc4096_FETCH_SIZE")
    write_big, _ = counters("pmc4096_WRITE_SIZE")
    inst8, l8 = counters("pmc_inst_g8")
    dom_pmc = {"kernel": dom, "frames": frames}
    big = [k for k in fetch_big if dom in k and DOMINANT in k and k in write_big]
    fetch_g8, lg = counters("pmc_g8_FETCH_SIZE")
    write_g8, _ = counters("pmc_g8_WRITE_SIZE")
    small = [k for k in fetch_g8 if dom in k and DOMINANT in k and k in write_g8]
    if big:
        k = big[0]
        per_launch = (2 * fetch_big[k]["FETCH_SIZE"] + write_big[k]["WRITE_SIZE"]) * 1024 / max(1, lb[k])
        dom_pmc.update({"instantiation": k, "hbm_bytes_per_frame": per_launch / frames, "launches_seen": lb[k],
                        "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `python bench.py --distinct 64 --steps 1 --warmup 0` "
                                  f"({frames} frames in flight, 64 distinct ones repeated: rocprofv3 --pmc does not survive torch "
                                  f"synthesising {frames} frames one by one), (2*FETCH_SIZE + WRITE_SIZE) * 1024 per launch, "
                                  f"profiles/{TAG}_dominant_kernel_pmc.json"})
    if small:
        k = small[0]
        per_frame = (2 * fetch_g8[k]["FETCH_SIZE"] + write_g8[k]["WRITE_SIZE"]) * 1024 / max(1, lg[k]) / PMC_FRAMES
        dom_pmc["hbm_bytes_per_frame_at_64_frames"] = per_frame
        if not big:  # the 4096-frame PMC pass did not survive: the same instantiation with 64 frames stands in, and says so
            dom_pmc.update({"instantiation": k, "hbm_bytes_per_frame": per_frame, "frames": frames, "measured_with_frames": PMC_FRAMES,
                            "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `{DOMINANT_KNOBS} python bench.py --frames "
                                      f"{PMC_FRAMES} --steps 1 --warmup 0` (the bench's instantiation, with "
                                      f"{PMC_FRAMES} frames: the PMC pass with {frames} frames crashed in rocprofv3), per frame"})
    g8 = [k for k in inst8 if dom in k and DOMINANT in k]
    if g8:
        k = g8[0]
        samples = PMC_FRAMES * 4096 * 4096 * max(1, l8[k])
        dom_pmc.update({"valu_wave_instructions_per_sample": round(inst8[k]["SQ_INSTS_VALU"] / samples, 3),
                        "salu_wave_instructions_per_sample": round(inst8[k]["SQ_INSTS_SALU"] / samples, 3),
                        "lds_wave_instructions_per_sample": round(inst8[k]["SQ_INSTS_LDS"] / samples, 3),
                        "instruction_source": f"rocprofv3 --pmc SQ_INSTS_* of `{DOMINANT_KNOBS} python bench.py --frames 64 --steps 1 "
                                              "--warmup 0` (the instantiation of the 4096-frame launch: four scans per wavefront, four wavefronts "
                                              "per workgroup; per-sample counts do not depend on the number of wavefronts)"})
    with open(os.path.join(DST, f"{TAG}_dominant_kernel_pmc.json"), "w") as f:
        json.dump(dom_pmc, f, indent=1)
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
            if c.get("SQ_INSTS_VALU") and li.get(k):
                f.write(f"    per sample: VALU {c['SQ_INSTS_VALU']/samples:.2f}  SALU {c['SQ_INSTS_SALU']/samples:.2f}  "
                        f"LDS {c['SQ_INSTS_LDS']/samples:.3f}  wave-cycles(x4) {4*c['SQ_WAVE_CYCLES']/samples:.1f}\n")
    # ---- the rest of the collection run, as it is
    import shutil
    for src, dst in (("other_configs.txt", "other_configs.txt"), ("one_frame_latency.txt", "one_frame_latency.txt"),
                     ("copy_probe.txt", "copy_probe.txt"), ("pytest_gpu.log", "pytest_gpu.log"), ("decode_tulips.txt", "decode_natural_image_4096_frames.txt"),
                     ("bench_cfg4.json", "bench_cfg4_1gpu.json")):
        if os.path.exists(os.path.join(SRC, src)):
            shutil.copyfile(os.path.join(SRC, src), os.path.join(DST, f"{TAG}_{dst}"))
    one = os.path.join(SRC, "one", "one_kernel_stats.csv")
    if os.path.exists(one):
        with open(one, newline="") as f, open(os.path.join(DST, f"{TAG}_rocprof_kernel_stats_one_frame.csv"), "w") as g:
            g.write("# rocprofv3 --kernel-trace --stats -- python tools/one_frame_latency.py --calls 6 (ONE 4096 x 4096 frame through the host-pointer C ABI, 7 calls)\n")
            g.write("kernel,calls,total_ms,avg_us\n")
            for r in csv.DictReader(f):
                if "jls" in r["Name"]:
                    g.write(f"\"{short(r['Name'])}\",{r['Calls']},{float(r['TotalDurationNs'])/1e6:.3f},{float(r['AverageNs'])/1e3:.1f}\n")
    print("profiles written for", TAG, "value", bench["value"])


if __name__ == "__main__":
    main()
