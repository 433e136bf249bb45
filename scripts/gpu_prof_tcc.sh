#!/bin/bash
# Are the L2 <-> memory request counters device-wide?  HBM stream kernels in another process, the probe here.
set -u
mkdir -p gpurun_out
export GPUMON_PROF_EXTRA="${1:-TCC_EA0_RDREQ_sum,TCC_EA0_WRREQ_sum,TCC_EA0_RDREQ_32B_sum}"
timeout -k 5 60 build/bin/gpumon-diag hbm --bytes 4G --iters 400 > gpurun_out/prof_tcc_hbm.log 2>&1 &
load=$!
sleep 1.5
timeout -k 5 30 build/bin/gpumon-prof-probe --interval-ms 500 --count 4 > gpurun_out/prof_tcc.log 2>&1; rc=$?
wait $load; lrc=$?
echo "rc $rc $lrc"; tail -6 gpurun_out/prof_tcc.log; tail -8 gpurun_out/prof_tcc_hbm.log
[ $rc -eq 0 ]
