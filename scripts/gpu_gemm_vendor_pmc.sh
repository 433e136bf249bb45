#!/usr/bin/env bash
# Instruction mix, wait cycles and L2 traffic of this repository's bf16 GEMM kernels next to hipBLASLt's, same
# operands (8192^3), one rocprofv3 counter pass per run (rocprofv3 does not multiplex).  Run on the GPU box:
#   scripts/gpu_gemm_vendor_pmc.sh        -> gpurun_out/vendor_pmc/*.md
set -u
export TMPDIR=/tmp GEMM_VC_REPS=1 GEMM_VC_LAUNCHES=12
OUT="$PWD/gpurun_out/vendor_pmc"
mkdir -p "$OUT"
pass() {
  local name=$1; shift
  echo "=== pass $name: $*"
  timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$name" -o p -- python3 scripts/gpu_gemm_vendor_clock.py \
    > "$OUT/$name.log" 2>&1 || { echo "pass $name failed ($?)"; tail -5 "$OUT/$name.log"; exit 1; }
  python3 scripts/prof_summary.py "$OUT/$name" "$OUT/$name.md" "bf16 GEMM 8192^3, ours and hipBLASLt: $name" > /dev/null || exit 1
  grep -E "gemm_|Cijk_|^\| kernel" "$OUT/$name.md" | cut -c1-260 | head -8
}
pass insts SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES &&
pass waits SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL &&
pass l2rd FETCH_SIZE &&
pass l2wr WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
