"""rocprofv3 PMC output of the headline decoder -> effective clock, wavefront residency and how busy the vector ALUs were.

    cd /tmp && rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU \\
        SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-include-regex decode_scans --output-format csv -d OUT -o p -- \\
        python tools/decode_sweep.py --frames 4096 --distinct 32 --groups 16:4,32:8,8 --sizes 4096 --repeat 0
    python tools/pmc_decode_clocks.py OUT        (profiles/r05_pmc_decode_valu_busy.txt, r05_pmc_decode_effective_clock.txt)

GRBM_GUI_ACTIVE is summed over the 8 XCDs (effective clock = GRBM_GUI_ACTIVE / 8 / kernel time, MI355X_MICROARCH.md "DVFS give-back");
the SQ_* counters count quad-cycles, SQ_BUSY_CYCLES per shader engine (32 of them)."""
import collections
import csv
import os
import sys

base = sys.argv[1]
rows = list(csv.DictReader(open(os.path.join(base, "p_counter_collection.csv"))))
trace = {r["Dispatch_Id"]: r for r in csv.DictReader(open(os.path.join(base, "p_kernel_trace.csv")))}
launches = collections.OrderedDict()
for r in rows:
    a = launches.setdefault(r["Dispatch_Id"], {})
    a[r["Counter_Name"]] = float(r["Counter_Value"])
    a["name"] = r["Kernel_Name"].split("(")[0].replace("void jls::", "")
    a["grid"], a["wg"] = int(r["Grid_Size"]), int(r["Workgroup_Size"])
for d, v in launches.items():
    k = trace[d]
    dur = (int(k["End_Timestamp"]) - int(k["Start_Timestamp"])) / 1e9
    waves = v["grid"] // 64
    clock = v["GRBM_GUI_ACTIVE"] / 8 / dur
    simd_cycles = 1024 * clock * dur
    print(f"{v['name']}: workgroups of {v['wg']} threads, {waves} wavefronts ({waves / 256:.0f} per CU), kernel {dur:.3f} s, effective clock {clock / 1e9:.3f} GHz")
    if "SQ_WAVE_CYCLES" in v:
        resident = v["SQ_WAVE_CYCLES"] * 4 / waves / clock
        print(f"    a wavefront is resident for {resident:.2f} s on average = {100 * resident / dur:.0f} % of the kernel")
    if "SQ_BUSY_CYCLES" in v:
        print(f"    shader engines busy {100 * v['SQ_BUSY_CYCLES'] / 32 / (clock * dur):.0f} % of the kernel")
    if "SQ_ACTIVE_INST_VALU" in v and "SQ_INSTS_VALU" in v:
        print(f"    SQ_INSTS_VALU {v['SQ_INSTS_VALU']:.4e}, SQ_ACTIVE_INST_VALU {v['SQ_ACTIVE_INST_VALU']:.4e} quad-cycles: vector ALU busy "
              f"{400 * v['SQ_ACTIVE_INST_VALU'] / simd_cycles:.1f} % of all SIMD cycles; a vector instruction every {simd_cycles / v['SQ_INSTS_VALU']:.2f} SIMD cycles")
    for name, label in (("SQ_ACTIVE_INST_ANY", "any instruction active"), ("SQ_ACTIVE_INST_LDS", "LDS"), ("SQ_INST_CYCLES_SALU", "scalar")):
        if name in v:
            print(f"    {label}: {400 * v[name] / simd_cycles:.1f} % of the SIMD cycles")
