#!/bin/bash
# Which thread of a process that has the counter plugin open uses CPU, and in which system call.
set -u
mkdir -p gpurun_out
timeout -k 5 30 build/bin/gpumon-prof-probe --interval-ms 1000 --count 6 > gpurun_out/prof_threads_probe.log 2>&1 &
w=$!
sleep 3
pid=$(pgrep -P $w | head -1)
[ -n "$pid" ] || pid=$w
for rep in 1 2 3; do
  for t in /proc/$pid/task/*; do
    echo "$(basename $t) $(awk '{print "utime", $14, "stime", $15}' $t/stat) syscall $(cut -d' ' -f1-3 $t/syscall 2>/dev/null) wchan $(cat $t/wchan 2>/dev/null)"
  done
  echo --
  sleep 1
done > gpurun_out/prof_threads.log 2>&1
wait $w
cat gpurun_out/prof_threads.log
