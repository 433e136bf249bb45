#!/bin/bash
# CPU cost of the counter plugin: per-call wall and CPU time on an idle device and while a GEMM
# soak runs in another process.
set -u
mkdir -p gpurun_out
timeout -k 5 30 build/bin/gpumon-prof-probe --interval-ms 1000 --count 3 > gpurun_out/prof_cost_idle.log 2>&1; r0=$?
timeout -k 5 60 build/bin/gpumon-diag gemm --dtype bf16 --seconds 8 > gpurun_out/prof_cost_gemm.log 2>&1 &
gemm=$!
sleep 2
timeout -k 5 30 build/bin/gpumon-prof-probe --interval-ms 1000 --count 3 > gpurun_out/prof_cost_load.log 2>&1; r1=$?
wait $gemm; grc=$?
echo "rc $r0 $r1 $grc"
tail -5 gpurun_out/prof_cost_idle.log; tail -5 gpurun_out/prof_cost_load.log
[ $r0 -eq 0 ] && [ $r1 -eq 0 ] && [ $grc -eq 0 ]
