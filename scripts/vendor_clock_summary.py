#!/usr/bin/env python3
"""Turns the rocprofv3 output of scripts/gpu_gemm_vendor_clock.py (one GRBM_GUI_ACTIVE pass) into the
table of clock, time and cycles per GEMM kernel: effective clock = GRBM_GUI_ACTIVE / 8 / duration
(the counter sums the eight XCDs); medians over the last two thirds of each kernel's dispatches.

usage: vendor_clock_summary.py <rocprof-out-dir> <out.md>
"""
import csv
import glob
import os
import re
import statistics
import sys
from collections import defaultdict


def main():
    src, out = sys.argv[1], sys.argv[2]
    per = defaultdict(list)  # kernel -> [(dur_us, gui_active)]
    for path in glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != "GRBM_GUI_ACTIVE":
                    continue
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
                per[row["Kernel_Name"]].append((dur, float(row["Counter_Value"])))
    rows = []
    for name, xs in per.items():
        if len(xs) < 30 or not re.search(r"gemm_|Cijk_", name):
            continue
        xs = xs[len(xs) // 3:]
        dur = statistics.median(d for d, _ in xs)
        cyc = statistics.median(c for _, c in xs) / 8
        label = re.sub(r"\(anonymous namespace\)::|^void ", "", name).split("(")[0]
        rows.append((dur, label, cyc / dur, cyc, len(xs)))
    rows.sort()
    lines = ["# bf16 GEMM 8192^3: this repository's kernels and hipBLASLt's in one rocprofv3 counter pass", "",
             "`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -- python3 scripts/gpu_gemm_vendor_clock.py`; effective clock = "
             "GRBM_GUI_ACTIVE / 8 / duration; medians over the last two thirds of each kernel's dispatches; sorted by time.", "",
             "| kernel | dispatches | clock MHz | time us | cycles |", "|---|---:|---:|---:|---:|"]
    for dur, label, mhz, cyc, n in rows:
        lines.append(f"| `{label}` | {n} | {mhz:.0f} | {dur:.1f} | {cyc:.0f} |")
    lines += ["", "The matrix pipe needs 1,048,576 cycles for this product (128 K-tiles x 4 tiles per CU x 2048)."]
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
