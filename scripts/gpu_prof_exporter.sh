#!/bin/bash
# The exporter with --prof-counters on a real device while a bf16 GEMM soak runs in another
# process: dcgm_fi_prof_pipe_tensor_active must follow the load.  Output under gpurun_out/.
set -u
mkdir -p gpurun_out
out=gpurun_out/prof_exporter
rm -f $out.*.prom
timeout -k 5 90 build/bin/gpumon-exporter -e --prof-counters --self-metrics -o $out.prom -d 500 2> $out.log &
exp=$!
sleep 4
cp $out.prom $out.idle.prom 2>/dev/null
timeout -k 5 60 build/bin/gpumon-diag gemm --dtype bf16 --seconds 5 > $out.gemm.log 2>&1 &
gemm=$!
sleep 3.5
cp $out.prom $out.load.prom 2>/dev/null
grep -E "VmRSS|Threads" /proc/$exp/status > $out.status
cat /proc/$exp/stat | awk '{print "utime_ticks", $14, "stime_ticks", $15}' >> $out.status
wait $gemm; grc=$?
sleep 2
cp $out.prom $out.after.prom 2>/dev/null
kill -TERM $exp; wait $exp; erc=$?
echo "exporter rc=$erc gemm rc=$grc"
for f in idle load after; do echo "== $f"; grep -E "^dcgm_fi_prof_(pipe_tensor_active|gr_engine_active|dram_active)" $out.$f.prom; done
cat $out.status; tail -5 $out.log
[ $grc -eq 0 ]
