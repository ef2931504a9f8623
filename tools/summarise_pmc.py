#!/usr/bin/env python3
"""Per-kernel sums of the counters of one rocprofv3 --pmc run: python tools/summarise_pmc.py <dir with *_counter_collection.csv>"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"(jls::[\w:]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:60]


def main(root):
    files = glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for path in files:
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = short(row["Kernel_Name"])
                if not k.startswith("jls::"):
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row["Dispatch_Id"])
    names = sorted({c for k in acc for c in acc[k]})
    print("kernel,launches," + ",".join(names))
    for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0)):
        print(f"\"{k}\",{len(launches[k])}," + ",".join(f"{acc[k].get(c, 0):.0f}" for c in names))


if __name__ == "__main__":
    main(sys.argv[1])
