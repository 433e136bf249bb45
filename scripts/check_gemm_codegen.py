#!/usr/bin/env python3
"""Reads the disassembly of the four-wave GEMM kernels in a built libgpumon_diag.so and checks what the source cannot
express: the MFMAs are inline asm, so the compiler knows nothing of their result latency, and the copies out of the
accumulator registers that the epilogue needs (v_accvgpr_read) are the compiler's to place.  Every such copy must have
a matrix-pipe drain (s_nop 15) between it and the last MFMA before it in program order; the K loop must hold no
scratch access and no s_waitcnt vmcnt that the source did not write (a spill reload there would drain the DMA queue).

usage: check_gemm_codegen.py [lib.so]     exit status 0 = all kernels pass; prints one line per kernel
"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def disassemble(lib):
    with tempfile.TemporaryDirectory() as tmp:
        local = os.path.join(tmp, "lib.so")  # the extracted code objects are written next to the library
        shutil.copy(lib, local)
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=tmp, check=True, capture_output=True)
        objs = [os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f]
        if not objs:
            raise RuntimeError(f"no gfx950 code object in {lib}")
        out = "".join(subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", o], check=True, capture_output=True, text=True).stdout
                      for o in sorted(objs))  # one code object per translation unit
    kernels, name = {}, None
    for line in out.splitlines():
        m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
        if m:
            name = m.group(1)
            kernels[name] = []
        elif name and line.startswith("\t"):
            kernels[name].append(line.split("//")[0].strip())
    return kernels


def check(name, ins):
    problems = []
    mfma = [i for i, l in enumerate(ins) if l.startswith("v_mfma")]
    if not mfma:
        return ["no MFMA found"]
    drains = [i for i, l in enumerate(ins) if l.startswith("s_nop 15")]
    early = 0
    for i, l in enumerate(ins):
        if l.startswith("v_accvgpr_read") and i > mfma[0]:
            last = max(x for x in mfma if x < i)
            if not any(last < d < i for d in drains):
                early += 1
    if early:
        problems.append(f"{early} accumulator reads with no drain since the last MFMA")
    loop = ins[mfma[0]:mfma[-1] + 1]
    if any(l.startswith("scratch_") for l in loop):
        problems.append("scratch access between the first and the last MFMA")
    return problems


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "build", "libgpumon_diag.so")
    kernels = disassemble(lib)
    demangled = subprocess.run(["c++filt"], input="\n".join(kernels), capture_output=True, text=True).stdout.splitlines()
    bad = 0
    for mangled, pretty in zip(kernels, demangled):
        if "gemm_w4_kernel" not in pretty:
            continue
        short = re.sub(r"\(anonymous namespace\)::|void |\(.*", "", pretty)
        problems = check(short, kernels[mangled])
        print(f"{short}: {'ok' if not problems else '; '.join(problems)}")
        bad += bool(problems)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
