#!/bin/bash
# Decoded-vs-raw metrics table pairs, idle and under a GEMM soak (distinct values make the match unambiguous).
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
timeout -k 5 60 build/bin/metrics-layout-probe /sys/class/drm/card0/device/gpu_metrics 2 > gpurun_out/metrics_layout_idle.json || exit 1
timeout -k 10 40 build/bin/gpumon-diag gemm --dtype bf16 --seconds 8 --no-telemetry --no-expect > gpurun_out/metrics_layout_load.log 2>&1 &
LOAD=$!
sleep 3
timeout -k 5 60 build/bin/metrics-layout-probe /sys/class/drm/card0/device/gpu_metrics 4 > gpurun_out/metrics_layout_load.json
RC=$?
wait $LOAD
exit $RC
