#!/usr/bin/env python3
"""Reads the disassembly of the four-wave GEMM kernels in a built libgpumon_diag.so and checks what the source cannot
express: the MFMAs are inline asm, so the compiler knows nothing of their result latency, and the copies out of the
accumulator registers that the epilogue needs (v_accvgpr_read) are the compiler's to place.  Every such copy must have
a matrix-pipe drain (s_nop 15) between it and the last MFMA before it in program order; the K loop must hold no
scratch access (a spill reload there would drain the DMA queue) and no write to M0 between the four LDS-DMA pieces of
a half-tile (only the first of them tells the compiler about M0).

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
    # M0 is the LDS base of the LDS-DMA pieces: one write per half-tile (the s_mov_b32 of piece 0) and per MX scale unit,
    # each followed by the loads that use it; pieces 1..3 do not declare M0, so a write of the compiler's own between a
    # piece 0 and its piece 3 would send them elsewhere.
    dma = [i for i, l in enumerate(loop) if l.startswith("global_load_lds_dwordx4")]
    m0 = [i for i, l in enumerate(loop) if re.match(r"s_\w+ m0\b", l)]
    if len(dma) % 4:
        problems.append("LDS-DMA pieces do not come in fours")
    for g in range(0, len(dma) - 3, 4):
        inside = [i for i in m0 if dma[g] < i < dma[g + 3]]
        if inside:
            problems.append(f"M0 written between the pieces of one half-tile (instruction {inside[0]} of the loop)")
            break
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
