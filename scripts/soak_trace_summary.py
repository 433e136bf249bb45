"""Kernel trace of a GEMM soak: duration of the GEMM and checksum kernels and the idle gaps before each of them.
usage: soak_trace_summary.py <rocprofv3 output dir> [out.md]"""
import csv
import glob
import statistics
import sys

files = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 4: 3 * len(rows) // 4]   # steady state
by = {}
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:48]
    d = by.setdefault(name, {"dur": [], "gap": []})
    d["dur"].append((e - s) / 1e3)
    if prev_end is not None:
        d["gap"].append((s - prev_end) / 1e3)
    prev_end = e
lines = ["| kernel | launches | median us | idle before it, median us |", "|---|---|---|---|"]
for name, d in by.items():
    lines.append(f"| {name} | {len(d['dur'])} | {statistics.median(d['dur']):.1f} | {statistics.median(d['gap']) if d['gap'] else 0:.1f} |")
text = "\n".join(lines)
print(text)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(text + "\n")
