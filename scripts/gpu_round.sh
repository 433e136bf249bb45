#!/bin/bash
# One GPU-box session: GPU tests, smoke, bench, diag CLI, profiles.  Every step runs
# under its own time limit; a step that ends with anything but 0 (ok) or 1 (an
# ordinary test failure) ends the session, so nothing is launched after a fault,
# an abort or a hang.
# usage: scripts/gpu_round.sh [step ...]   (default: tests smoke bench diag)
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out
mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0

step() {  # step <name> <seconds> <cmd...>
  local name=$1 secs=$2
  shift 2
  echo "=== $name ($(date +%T))"
  timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "=== $name exit $rc"
  tail -n 15 "$OUT/$name.log"
  if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then
    echo "stopping: $name ended with $rc"
    exit $rc
  fi
}

STEPS=${*:-tests smoke bench diag}
for s in $STEPS; do
  case $s in
    tests) step gpu_tests 900 python -u -m pytest tests -m gpu -v --timeout 120 --timeout-method thread ;;
    smoke) step smoke 300 python -u -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) step bench 600 python -u bench.py --steps 50 --warmup 5 ;;
    diag)
      step diag_run2 600 build/bin/gpumon-diag run -r 2
      step diag_rccl 300 build/bin/gpumon-diag rccl --max-bytes 256M --factor 16
      step diag_links 120 build/bin/gpumon-diag links
      ;;
    soak)
      # the long suite on one GPU: whole-HBM sweep with march, 60 s soak with the matrix-pipe counter next to it
      step diag_run3 900 build/bin/gpumon-diag run -r 3 --mfma-counters --publish "$OUT/diag_run3_report.json"
      ;;
    scrape)
      # exporter cost at 1 s resolution: real device library, then the same binary while the GPU soaks
      step scrape_amdsmi 120 python -u scripts/scrape_bench.py --backend amdsmi --seconds 30 --out "$OUT/scrape_amdsmi.json"
      ;;
    scrape8)
      # an 8-GPU node's sampling cost on the real library: every device presented eight times, 1 s and 100 ms resolution
      step scrape_amdsmi_x8 150 python -u scripts/scrape_bench.py --backend amdsmi --replicate 8 --seconds 62 --out "$OUT/scrape_amdsmi_x8.json"
      step scrape_amdsmi_x8_100ms 150 python -u scripts/scrape_bench.py --backend amdsmi --replicate 8 --seconds 62 --interval-ms 100 --out "$OUT/scrape_amdsmi_x8_100ms.json"
      ;;
    vendorclock)
      export TMPDIR=/tmp
      step vendor_clock 400 timeout -s KILL 380 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$PWD/$OUT/vendor_clock" -o v -- python3 scripts/gpu_gemm_vendor_clock.py
      step vendor_clock_summary 60 python scripts/vendor_clock_summary.py "$OUT/vendor_clock" "$OUT/gemm_vendor_clock.md"
      ;;
    probe) step probe 300 python -u scripts/gpu_probe.py ;;
    copyref) step copy_ref 300 python -u scripts/gpu_copy_ref.py ;;
    prof)
      # kernel trace of a short bench run; rocprofv3 wants a writable cwd-independent TMPDIR
      export TMPDIR=/tmp
      step prof_bench 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$PWD/$OUT/prof_bench" -o bench -- python3 bench.py --steps 5 --warmup 2 --no-exporter
      python scripts/prof_summary.py "$OUT/prof_bench" "$OUT/bench_kernel_trace.md" "bench.py --steps 5 --warmup 2: kernel trace" > /dev/null
      ;;
    pmc)
      # One counter pass per run (rocprofv3 does not multiplex): SQ pass, L2 read pass, L2 write/hit pass.
      export TMPDIR=/tmp
      D="$PWD/build/bin/gpumon-diag"
      for target in "gemm_bf16:gemm --dtype bf16 --seconds 0.05 --no-telemetry --no-expect" "gemm_fp8:gemm --dtype fp8 --seconds 0.05 --no-telemetry --no-expect" \
                    "gemm_mxfp8:gemm --dtype mxfp8 --seconds 0.05 --no-telemetry --no-expect" \
                    "gemm_split:gemm --dtype bf16 --m 2048 --n 2048 --k 16384 --seconds 0.05 --no-telemetry --no-expect" \
                    "hbm:hbm --bytes 2G --iters 2 --no-expect" "memtest:memtest --bytes 4G --chunk 4G --patterns random,address --march --no-expect"; do
        name=${target%%:*}; args=${target#*:}
        buf=""  # buffer size of the streaming targets, for the expected-bytes columns
        [ "$name" = hbm ] && buf=2147483648
        [ "$name" = memtest ] && buf=4294967296
        case " ${PMC_TARGETS:-gemm_bf16 gemm_fp8 gemm_mxfp8 gemm_split hbm memtest} " in *" $name "*) ;; *) continue ;; esac
        step "pmc_sq_$name" 120 timeout -s KILL 100 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_UNALIGNED_STALL \
          --output-format csv -d "$PWD/$OUT/pmc_sq_$name" -o p -- "$D" $args
        step "pmc_rd_$name" 120 timeout -s KILL 100 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$PWD/$OUT/pmc_rd_$name" -o p -- "$D" $args
        step "pmc_wr_$name" 120 timeout -s KILL 100 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$PWD/$OUT/pmc_wr_$name" -o p -- "$D" $args
        for pass in sq rd wr; do
          python scripts/prof_summary.py "$OUT/pmc_${pass}_$name" "$OUT/pmc_${pass}_$name.md" "$pass counters: gpumon-diag $name" $buf > /dev/null
        done
      done
      ;;
    *) echo "unknown step $s"; exit 2 ;;
  esac
done
echo "=== session complete"
