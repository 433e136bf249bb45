#!/bin/bash
# Runs the diagnostics on every GPU the container (or the host) sees and leaves the reports in $GPUMON_DIAG_DIR for
# `gpumon-exporter --diag-results DIR`.  All GPUs are loaded at the same time, step by step (one process, one host
# thread per GPU): that is what the node's power delivery and cooling have to carry in production, and every GPU is
# also held against the median of the others.  The fabric tests (xGMI link test, RCCL sweep) run once over all GPUs,
# not once per GPU.  Exit status: 0 when every test passed (warnings included), 1 when a test failed, 2 when the
# diagnostics could not run or a step died (nothing is started after a step that was killed).
#   GPUMON_DIAG_LEVEL       1 (seconds), 2 (about a minute, default), 3 (whole-HBM sweep, 60 s soak, fabric)
#   GPUMON_DIAG_DIR         report directory (default /var/lib/gpumon/diag)
#   GPUMON_DIAG_SUBMIT      address of a host engine that also gets the verdicts (e.g. 127.0.0.1:5555); empty = none
#   GPUMON_DIAG_SEQUENTIAL  1 = one GPU after the other (one report per GPU, no comparison between GPUs)
#   GPUMON_DIAG_EXPECT_GPUS number of GPUs the node should have (the deployment checks fail on another count)
set -u
LEVEL=${GPUMON_DIAG_LEVEL:-2}
DIR=${GPUMON_DIAG_DIR:-/var/lib/gpumon/diag}
SUBMIT=${GPUMON_DIAG_SUBMIT:-}
DIAG=${GPUMON_DIAG_BIN:-gpumon-diag}
mkdir -p "$DIR" || exit 2
info=$("$DIAG" info) || { echo "gpumon-diag-all: no GPU to test" >&2; exit 2; }
n=$(printf '%s\n' "$info" | grep -c '^GPU ')
[ "$n" -gt 0 ] || { echo "gpumon-diag-all: no GPU to test" >&2; exit 2; }
extra=()
[ -n "$SUBMIT" ] && extra=(--submit "$SUBMIT")
worst=0
note() {  # keep the worst status: 2 (could not run) over 1 (failed) over 0
  if [ "$1" -ge 124 ]; then
    # killed by a signal or a time limit (abort, segmentation fault, GPU fault): nothing more is started on the
    # GPUs of this node, the operator reads what the run left
    echo "gpumon-diag-all: the last step ended with status $1; stopping" >&2
    exit 2
  fi
  [ "$1" -gt "$worst" ] && worst=$1
  return 0
}
echo "=== deployment checks"
"$DIAG" preflight ${GPUMON_DIAG_EXPECT_GPUS:+--expect-gpus "$GPUMON_DIAG_EXPECT_GPUS"} --publish "$DIR/preflight.json" "${extra[@]}"
note $?
if [ "${GPUMON_DIAG_SEQUENTIAL:-0}" = 1 ]; then
  rm -f "$DIR/node.json"  # the report of an earlier concurrent run would contradict the ones written now
  for ((g = 0; g < n; g++)); do
    echo "=== GPU $g: level $LEVEL"
    "$DIAG" run -r "$LEVEL" -g "$g" --no-fabric --publish "$DIR/gpu$g.json" "${extra[@]}"
    note $?
  done
else
  rm -f "$DIR"/gpu[0-9].json "$DIR"/gpu[0-9][0-9].json  # per-GPU reports of an earlier sequential run
  echo "=== $n GPU(s) at once: level $LEVEL"
  "$DIAG" run -r "$LEVEL" --gpus all --no-fabric --publish "$DIR/node.json" "${extra[@]}"
  note $?
fi
if [ "$n" -gt 1 ]; then
  echo "=== xGMI links"
  "$DIAG" links --publish "$DIR/links.json" "${extra[@]}"
  note $?
  echo "=== RCCL"
  if [ "$LEVEL" -ge 3 ]; then
    "$DIAG" rccl --scaling --publish "$DIR/rccl.json" "${extra[@]}"
  else
    "$DIAG" rccl --max-bytes 256M --factor 16 --publish "$DIR/rccl.json" "${extra[@]}"
  fi
  note $?
fi
exit "$worst"
