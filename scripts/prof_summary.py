#!/usr/bin/env python3
"""Condenses rocprofv3 CSV output (kernel trace and/or counter collection) into a
small markdown report: per-kernel call counts and times, per-kernel counter sums.

The register and LDS table comes from the code objects of the in-tree libraries
(scripts/kernel_resources.py) where the kernel is one of ours: rocprofv3's own
columns show 0 bytes of LDS for LDS requested at launch and no accumulator
registers.  Kernels of other libraries keep rocprofv3's figures, marked as such.

FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes, so a wide coalesced
stream (16 bytes per lane, every kernel here) shows HALF the bytes it fetched; WRITE_SIZE is exact for 16-byte-per-lane
stores.  With a buffer size the report adds the bytes those counters stand for next to what the kernel is known to
move (reads / writes as multiples of the buffer: copy 1 / 1, triad 2 / 1, ...).

usage: prof_summary.py <dir-with-csv> <out.md> [title] [buffer-bytes]
"""
import csv
import glob
import os
import re
import sys
from collections import defaultdict


def short(name, n=72):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0] if len(name) > n else name
    return name if len(name) <= n else name[: n - 3] + "..."


def code_object_table():
    try:
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import kernel_resources
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        libs = [os.path.join(root, "build", n) for n in ("libgpumon_diag.so", "libgpumon_prof.so")]
        return kernel_resources.load([p for p in libs if os.path.exists(p)])
    except Exception as e:  # the summary is still useful without the table
        print(f"prof_summary: no code-object table ({e})", file=sys.stderr)
        return {}


# (reads, writes) of the streaming kernels as multiples of the buffer the command was given
TRAFFIC = [("onepass_kernel<0", 1, 1), ("onepass_kernel<1", 1, 1), ("onepass_kernel<2", 2, 1), ("onepass_kernel<3", 2, 1),
           ("onepass_kernel<5", 0, 1), ("read_kernel", 1, 0), ("fill_kernel", 0, 1), ("verify_kernel", 1, 0), ("march_kernel", 1, 1)]


PER_LANE = ("onepass_kernel", "fill_kernel", "march_kernel")  # one 16-byte access per work-item and operand: the grid says how many bytes


def expected_traffic(kernel, buffer_bytes, grid):
    for prefix, r, w in TRAFFIC:
        if kernel.startswith(prefix):
            n = 16 * grid if grid and kernel.startswith(PER_LANE) else buffer_bytes
            return r * n, w * n
    return None


def main():
    src, out = sys.argv[1], sys.argv[2]
    buffer_bytes = int(float(sys.argv[4])) if len(sys.argv) > 4 else 0
    ours = code_object_table()
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(src.rstrip("/"))
    lines = [f"# {title}", ""]
    for path in sorted(glob.glob(os.path.join(src, "**", "*kernel_trace.csv"), recursive=True)):
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3  # us
                a = agg[short(row["Kernel_Name"])]
                a[0] += 1
                a[1] += dur
                a[2] = min(a[2], dur)
                a[3] = max(a[3], dur)
        total = sum(a[1] for a in agg.values()) or 1.0
        lines += [f"## kernel trace ({os.path.basename(path)})", "", "| kernel | calls | total us | avg us | min us | max us | % |",
                  "|---|---:|---:|---:|---:|---:|---:|"]
        for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]:
            lines.append(f"| `{k}` | {a[0]} | {a[1]:.1f} | {a[1] / a[0]:.1f} | {a[2]:.1f} | {a[3]:.1f} | {100 * a[1] / total:.1f} |")
        lines.append("")
    for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
        agg = defaultdict(lambda: defaultdict(float))
        calls = defaultdict(set)
        regs = {}
        with open(path, newline="") as f:
            rows = list(csv.DictReader(f))
            # one kernel launched over different sizes (a 4 GiB chunk and the 512 MiB scrub buffer): one line per size
            grids = defaultdict(set)
            for row in rows:
                grids[short(row["Kernel_Name"])].add(row.get("Grid_Size", ""))
            for row in rows:
                k = short(row["Kernel_Name"])
                if len(grids[k]) > 1:
                    k += f" [grid {row.get('Grid_Size', '?')}]"
                agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
                calls[k].add(row.get("Dispatch_Id", ""))
                regs[k] = (row.get("VGPR_Count", "?"), row.get("Accum_VGPR_Count", "?"), row.get("SGPR_Count", "?"),
                           row.get("LDS_Block_Size", "?"), row.get("Workgroup_Size", "?"), row.get("Grid_Size", "?"))
        counters = sorted({c for v in agg.values() for c in v})
        lines += [f"## counters ({os.path.basename(path)})", "", "| kernel | dispatches | " + " | ".join(counters) + " |",
                  "|---|---:|" + "---:|" * len(counters)]
        for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values()))[:20]:
            n = max(1, len(calls[k]))
            lines.append(f"| `{k}` | {n} | " + " | ".join(f"{v.get(c, 0) / n:.4g}" for c in counters) + " |")
        lines += ["", "(counter values are per-dispatch means)", ""]
        if "FETCH_SIZE" in counters or "WRITE_SIZE" in counters:
            lines += ["FETCH_SIZE and WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes: a 16-byte-per-lane",
                      "stream shows half of what it fetched, so bytes read = 2 x FETCH_SIZE x 1024; WRITE_SIZE x 1024 is exact.", ""]
            if buffer_bytes:
                lines += [f"With {buffer_bytes / 2**30:.2f} GiB per buffer:", "",
                          "| kernel | bytes read (2 x FETCH_SIZE KiB) | expected | bytes written (WRITE_SIZE KiB) | expected |", "|---|---:|---:|---:|---:|"]
                for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1].values())):
                    try:
                        grid = int(regs[k][5])
                    except (KeyError, ValueError):
                        grid = 0
                    exp = expected_traffic(k, buffer_bytes, grid)
                    if not exp:
                        continue
                    n = max(1, len(calls[k]))
                    rd = f"{2 * 1024 * v['FETCH_SIZE'] / n / 2**30:.3f} GiB" if "FETCH_SIZE" in v else "-"
                    wr = f"{1024 * v['WRITE_SIZE'] / n / 2**30:.3f} GiB" if "WRITE_SIZE" in v else "-"
                    lines.append(f"| `{k}` | {rd} | {exp[0] / 2**30:.3f} GiB | {wr} | {exp[1] / 2**30:.3f} GiB |")
                lines.append("")
        lines += [
                  "| kernel | VGPR (all) | of which AGPR | SGPR | VGPR spills | LDS bytes (static + at launch) | workgroup | grid | source |",
                  "|---|---:|---:|---:|---:|---:|---:|---:|---|"]
        for k, r in regs.items():
            co = ours.get(k.split(" [grid")[0].split("(")[0])
            if co:
                lines.append(f"| `{k}` | {co['vgpr']} | {co['agpr']} | {co['sgpr']} | {co['spill_vgpr']} | "
                             f"{co['lds_static']} + {co['lds_dynamic']} | {r[4]} | {r[5]} | code object |")
            else:
                lines.append(f"| `{k}` | {r[0]} | {r[1]} | {r[2]} | ? | {r[3]} | {r[4]} | {r[5]} | rocprofv3 |")
        lines.append("")
    with open(out, "w") as f:
        f.write("\n".join(lines) + "\n")
    print(f"wrote {out} ({len(lines)} lines)")


if __name__ == "__main__":
    main()
