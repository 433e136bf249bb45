#!/usr/bin/env python3
"""Register, scratch and LDS use of every GPU kernel in a built library, read from the
code object's own metadata (the AMDGPU note record the loader uses), not from a profiler.

rocprofv3's VGPR/AGPR/LDS columns describe the dispatch packet and the granulated
register allocation: they show 0 bytes of LDS for a kernel whose LDS is requested at
launch and never show accumulator registers on a unified register file.  The code
object states what the compiler allocated: architected and accumulator VGPRs, spills,
scratch, static LDS.  LDS requested at launch is not in the code object; the table
below adds it for the kernels of this repository that are launched that way.

usage: kernel_resources.py [lib.so ...] [--json out.json] [--md out.md]
"""
import argparse
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"

# LDS requested at launch (bytes), by kernel-name pattern: the GEMM kernels take a 128 KiB operand ring,
# the four-wave MX-fp8 kernel (first template argument 3 = MX-fp8) 8 KiB more for its scale ring.
DYNAMIC_LDS = [
    (re.compile(r"gemm_w4_kernel<3,.*true>$"), 131072),  # unit scales: no scale ring
    (re.compile(r"gemm_w4_kernel<3,"), 131072 + 8192),
    (re.compile(r"gemm_s128_kernel<"), 163840),  # five stages of 32 KiB: all of a CU's LDS
    (re.compile(r"gemm_(pp|ps|w4)?_?kernel<"), 131072),
]


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout
    return out.splitlines()


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0]


def kernels_of(lib):
    """[{name, vgpr, agpr, sgpr, spill_vgpr, spill_sgpr, scratch, lds_static, lds_dynamic, max_workgroup}]"""
    tmp = tempfile.mkdtemp(prefix="kres-")
    try:
        local = os.path.join(tmp, "lib.so")
        shutil.copy(lib, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], capture_output=True, check=True, cwd=tmp)
        rows = []
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", os.path.join(tmp, f)], capture_output=True, text=True).stdout
            # One "- .agpr_count: ..." block per kernel, keys in alphabetical order inside amdhsa.kernels.
            for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                block = ".agpr_count:" + block
                kv = dict(re.findall(r"\.(\w+):\s+'?([^\n']+)'?", block))
                if "name" not in kv:
                    continue
                rows.append({
                    "mangled": kv["name"], "vgpr": int(kv.get("vgpr_count", 0)), "agpr": int(kv.get("agpr_count", 0)),
                    "sgpr": int(kv.get("sgpr_count", 0)), "spill_vgpr": int(kv.get("vgpr_spill_count", 0)),
                    "spill_sgpr": int(kv.get("sgpr_spill_count", 0)), "scratch": int(kv.get("private_segment_fixed_size", 0)),
                    "lds_static": int(kv.get("group_segment_fixed_size", 0)),
                    "max_workgroup": int(kv.get("max_flat_workgroup_size", 0)),
                })
        for r, name in zip(rows, demangle([r["mangled"] for r in rows])):
            r["name"] = short(name)
            r["lds_dynamic"] = next((b for pat, b in DYNAMIC_LDS if pat.search(r["name"])), 0)
            del r["mangled"]
        return rows
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load(paths):
    table = {}
    for lib in paths:
        for r in kernels_of(lib):
            table[r["name"]] = r
    return table


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="*", default=[os.path.join(ROOT, "build", n) for n in ("libgpumon_diag.so", "libgpumon_prof.so")])
    ap.add_argument("--json")
    ap.add_argument("--md")
    a = ap.parse_args()
    table = load([p for p in a.libs if os.path.exists(p)])
    if a.json:
        with open(a.json, "w") as f:
            json.dump(table, f, indent=1, sort_keys=True)
    lines = ["# Kernel resources from the code objects (gfx950)", "",
             "Architected and accumulator VGPRs share one 512-entry file per SIMD lane; the VGPR column is a wave's whole",
             "allocation, the AGPR column the part of it addressed as accumulators.",
             "LDS = static (in the code object) + requested at launch.", "",
             "| kernel | VGPR (all) | of which AGPR | SGPR | VGPR spills | scratch B | LDS static B | LDS at launch B | max workgroup |",
             "|---|---:|---:|---:|---:|---:|---:|---:|---:|"]
    for name, r in sorted(table.items()):
        lines.append(f"| `{name}` | {r['vgpr']} | {r['agpr']} | {r['sgpr']} | {r['spill_vgpr']} | {r['scratch']} | "
                     f"{r['lds_static']} | {r['lds_dynamic']} | {r['max_workgroup']} |")
    text = "\n".join(lines) + "\n"
    if a.md:
        with open(a.md, "w") as f:
            f.write(text)
    else:
        sys.stdout.write(text)
    return 0


if __name__ == "__main__":
    sys.exit(main())
