"""What the GEMM source cannot express is checked in the machine code of the built library (no GPU needed): the
four-wave kernel issues its MFMAs from inline asm, so the compiler knows nothing of their result latency and is free
to place the copies out of the accumulator registers (v_accvgpr_read) wherever it likes behind the K loop.  A copy
that comes before the matrix-pipe drain reads a register that an MFMA may still be writing: numerically wrong only
now and then, and only on hardware.  scripts/check_gemm_codegen.py disassembles every instantiation and counts."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "build", "libgpumon_diag.so")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(OBJDUMP)), reason="needs the built diagnostics library and llvm-objdump")
def test_no_accumulator_is_read_before_the_matrix_pipe_is_drained():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "check_gemm_codegen.py"), LIB], capture_output=True, text=True,
                       timeout=300)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert r.returncode == 0, r.stdout + r.stderr
    # three data types x (per tile, general, stream), the per-tile and the general kernels once more with the summing epilogue
    # of the soak, and all three with the epilogue of a GEMM split along K (fp32 accumulators to the workspace)
    assert len(lines) == 21 and all(ln.endswith(": ok") for ln in lines), r.stdout
    assert sum(", 0, 3, false, false" in ln for ln in lines) == 3 and sum(", 0, 3, false, true" in ln for ln in lines) == 3, r.stdout
    assert sum(", 0, 4, false, false" in ln for ln in lines) == 3, r.stdout


def test_the_checker_sees_a_read_that_comes_too_early():
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_gemm_codegen as cg
    good = ["v_mfma_f32_16x16x32_bf16 a[0:3], v[0:3], v[4:7], a[0:3]", "s_waitcnt vmcnt(0)", "s_nop 15", "s_nop 15",
            "v_accvgpr_read_b32 v0, a0", "global_store_dwordx4 v[2:3], v[4:7], off"]
    assert cg.check("k", good) == []
    early = [good[0], good[4]] + good[1:4] + [good[5]]
    assert any("no drain" in p for p in cg.check("k", early))
    spill = [good[0], "scratch_load_dword v1, off, off", good[0]] + good[1:]
    assert any("scratch" in p for p in cg.check("k", spill))
    assert cg.check("k", ["s_endpgm"]) == ["no MFMA found"]
