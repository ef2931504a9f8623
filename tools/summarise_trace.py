"""CHARLS_AMD_TRACE=1 lines (stderr of a program that uses the host-pointer ABI) -> where the calls' time went and the timeline of the
shared launches.    python tools/summarise_trace.py trace.err"""
import re
import statistics as st
import sys

rows = []
for line in open(sys.argv[1], errors="replace"):
    m = re.search(r"trace (\w+) begin=([\d.]+) total=([\d.]+) upload=([\d.]+) sync=([\d.]+) submit=([\d.]+) \(launch=([\d.]+) of (\d+) scans\) copy_out=([\d.]+)", line)
    if m:
        rows.append((m.group(1),) + tuple(float(x) for x in m.groups()[1:]))
if not rows:
    sys.exit("no trace lines")
t0 = min(r[1] for r in rows)
for kind in ("encode", "decode"):
    rs = sorted((r for r in rows if r[0] == kind), key=lambda r: r[1])
    if not rs:
        continue
    print(f"{kind}: {len(rs)} calls")
    for name, i in (("total", 2), ("upload", 3), ("submit", 5), ("copy_out", 8)):
        vals = [r[i] for r in rs]
        print("  %-9s min %9.1f  median %9.1f  max %9.1f ms" % (name, min(vals), st.median(vals), max(vals)))
print("launches in the order of their leaders' calls (kind, leader's call began at ms, ms inside the launch, scans):")
for r in sorted((r for r in rows if r[7] > 0), key=lambda r: r[1]):
    print(f"  {r[0]:6s} {r[1] - t0:9.0f} {r[6]:9.0f} {int(r[7]):5d}")
