#!/bin/bash
# Profiling-counter plugin on a real device: idle windows, then windows while a bf16 GEMM soak
# runs in another process (the counters are device-wide).  Output under gpurun_out/.
# $1: optional comma list of up to three extra SQ counters to show next to the fixed five.
set -u
mkdir -p gpurun_out
export GPUMON_PROF_EXTRA="${1:-}"
tag="${2:-}"
timeout -k 5 40 build/bin/gpumon-prof-probe --interval-ms 500 --count 3 > gpurun_out/prof_probe_idle$tag.log 2>&1
rc=$?
echo "idle rc=$rc"; tail -5 gpurun_out/prof_probe_idle$tag.log
[ $rc -eq 0 ] || exit $rc
timeout -k 5 60 build/bin/gpumon-diag gemm --dtype bf16 --seconds 6 > gpurun_out/prof_probe_gemm$tag.log 2>&1 &
gemm=$!
sleep 2
timeout -k 5 40 build/bin/gpumon-prof-probe --interval-ms 500 --count 6 > gpurun_out/prof_probe_load$tag.log 2>&1
rc=$?
wait $gemm
grc=$?
echo "load rc=$rc gemm rc=$grc"; tail -8 gpurun_out/prof_probe_load$tag.log; tail -3 gpurun_out/prof_probe_gemm$tag.log
[ $rc -eq 0 ] && [ $grc -eq 0 ]
