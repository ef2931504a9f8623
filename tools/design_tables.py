"""Prints DESIGN.md's section 6.1 - 6.3 tables from the files tools/summarise_profiles.py wrote under profiles/ (round given as rNN).
    python tools/design_tables.py r06"""
import csv
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(ROOT, "profiles")
b = json.load(open(os.path.join(P, f"{tag}_bench_default_1gpu.json")))
pmc = json.load(open(os.path.join(P, f"{tag}_dominant_kernel_pmc.json")))
cfg4 = json.load(open(os.path.join(P, f"{tag}_bench_cfg4_1gpu.json")))


def n(x, digits=0):
    s = f"{x:,.{digits}f}".replace(",", " ")
    return s


stats_ms = None
with open(os.path.join(P, f"{tag}_rocprof_kernel_stats_bench_default.csv")) as f:
    for row in csv.reader(l for l in f if not l.startswith("#")):
        if row and "decode_scans_group" in row[0]:
            stats_ms = float(row[3])
            break
r, ir, nat, near, cpu = b["roofline"], b["issue_roofline"], b["value_natural_image"], b.get("near_lossless", {}), b["cpu_baseline"]
print("### 6.1")
print(f"| `value` | **{n(b['value'])} MPix/s** ({b['ms_per_step'] / 1e3:.3f} s per step) |")
print(f"| encode / decode | {n(b['encode_mpix_s'])} / {n(b['decode_mpix_s'])} MPix/s |")
print(f"| dominant kernel | {n(r['kernel_ms_per_launch'])} ms by HIP events, {n(stats_ms)} ms by rocprofv3 --stats |")
print(f"| roofline | {r['achieved']:.1f} / {r['peak']:.0f} GB/s = {r['frac']:.4f}, traffic / algorithmic {r['traffic'] / r['algorithmic_bytes_per_launch']:.2f} |")
print(f"| instructions per sample | {pmc['valu_wave_instructions_per_sample']:.2f} vector, {pmc['salu_wave_instructions_per_sample']:.2f} scalar, "
      f"{pmc['lds_wave_instructions_per_sample']:.2f} LDS |")
print(f"| issue_roofline | {ir['frac']:.2f} of the four-wavefront ruler, {ir['frac_of_one_wavefront_per_simd']:.2f} of the one-wavefront ruler |")
print(f"| value_natural_image | {n(nat['value'])} (encode {n(nat['encode_mpix_s'])} / decode {n(nat['decode_mpix_s'])} = "
      f"{100 * nat['decode_mpix_s'] / b['decode_mpix_s']:.0f} % of the bench frames' decode) |")
if near:
    print(f"| near_lossless | encode {n(near['encode_mpix_s'])} / decode {n(near['decode_mpix_s'])} |")
print(f"| cpu_baseline | one thread {cpu['value']:.1f} MPix/s; {cpu['all_cores']['threads']} threads {n(cpu['all_cores']['value'])} |")
print("### 6.2")
print("| frames in flight | " + " | ".join(str(x["frames"]) for x in b["batch_sweep"]) + " |")
print("| encode, MPix/s | " + " | ".join(n(x["encode_mpix_s"]) for x in b["batch_sweep"]) + " |")
print("| decode, MPix/s | " + " | ".join(n(x["decode_mpix_s"], 1 if x["decode_mpix_s"] < 100 else 0) for x in b["batch_sweep"]) + " |")
print("by data: " + ", ".join(f"{x['data']} {n(x['encode_mpix_s'])} / {n(x['decode_mpix_s'])}" for x in b["data_sweep"]))
sf, t = b["single_frame_ms"], b["threads_abi"]
print(f"one frame: encode {sf['encode']} ms (CharLS {cpu['single_frame_ms']['encode']} ms), decode {sf['decode']} ms ({cpu['single_frame_ms']['decode']} ms)")
print(f"threads_abi: {t['threads']} threads {n(t['value'])} ({t['engine_counters']['calls']} calls in {t['engine_counters']['launches']} launches), "
      f"{t['more'][0]['threads']} threads {n(t['more'][0]['value'])}")
h = b["batch_api_host_buffers"]
print(f"batch API from pinned memory: encode {n(h['encode_mpix_s'])} - {n(h['more'][0]['encode_mpix_s'])}, decode {n(h['decode_mpix_s'])} / {n(h['more'][0]['decode_mpix_s'])}")
print(f"cfg4: {n(cfg4['value'])} round trip")
print("### 6.3")
for line in open(os.path.join(P, f"{tag}_other_configs.txt")):
    m = re.match(r"(.*?): .*?(\d+) frames, encode (\d+) MPix/s, decode (\d+) MPix/s", line)
    if m:
        print(f"| {m.group(1)} | {m.group(2)} | {n(int(m.group(3)))} | {n(int(m.group(4)))} |")
