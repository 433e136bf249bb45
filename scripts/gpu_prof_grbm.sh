#!/bin/bash
# Validation of the GRBM-derived activity figures against loads of known character, each load in ANOTHER
# process: idle, HBM stream (memory bound, no matrix work), bare MFMA loop (matrix pipe only, no memory
# traffic), GEMM (both).  For every load one `gpumon-prof-probe --once` with the default rotation
# (GUI_ACTIVE, SPI_BUSY, TA_BUSY) and one with other device-wide busy signals worth knowing
# (EA_BUSY, TC_BUSY, CP_BUSY); plus the CPU cost of one probe run and of an exporter in child mode.
# Any step that fails ends the script.
set -u
cd "$(dirname "$0")/.."
OUT=gpurun_out/prof_grbm
mkdir -p "$OUT"
P=build/bin/gpumon-prof-probe
D=build/bin/gpumon-diag

probe() {  # probe <tag>
  local TIMEFORMAT="probe_cost user %U sys %S wall %R"
  { time timeout -k 5 40 $P --once --interval-ms 300 > "$OUT/$1.default.log"; } 2> "$OUT/$1.default.err" || return 1
  GPUMON_PROF_GRBM=GRBM_EA_BUSY,GRBM_TC_BUSY,GRBM_CP_BUSY timeout -k 5 40 $P --interval-ms 300 --count 3 > "$OUT/$1.other.log" 2> "$OUT/$1.other.err" || return 1
  GPUMON_PROF_GRBM=GRBM_SPI_BUSY,GRBM_TA_BUSY,GRBM_GUI_ACTIVE timeout -k 5 40 $P --interval-ms 300 --count 3 > "$OUT/$1.raw.log" 2> "$OUT/$1.raw.err" || return 1
  echo "== $1"; grep '^sample' "$OUT/$1.default.log"; grep probe_cost "$OUT/$1.default.err"
  grep -h 'window' "$OUT/$1.other.log" "$OUT/$1.raw.log" | sed 's/  raw.*//'
}

under() {  # under <tag> <load command...>: run the load in the background, probe 2 s in, wait for the load
  local tag=$1; shift
  timeout -k 5 90 "$@" > "$OUT/$tag.load.log" 2>&1 &
  local load=$!
  sleep 2.5
  probe "$tag"; local rc=$?
  wait $load; local lrc=$?
  [ $rc -eq 0 ] && [ $lrc -eq 0 ] || { echo "stopping: $tag probe rc $rc load rc $lrc"; tail -3 "$OUT/$tag.load.log" "$OUT/$tag.default.err"; exit 1; }
}

probe idle || { echo "stopping: idle probe failed"; cat "$OUT/idle.default.err"; exit 1; }
under hbm   $D hbm --bytes 4G --iters 10000 --kernel copy
under mfma  $D mfma --dtype bf16 --seconds 14
under gemm  $D gemm --dtype bf16 --seconds 14 --no-telemetry

# exporter in child mode, 10 s period, for 45 s: CPU of the daemon itself and of its children
echo "== exporter --prof-counters=child"
GPUMON_BACKEND=amdsmi timeout -k 5 80 python -u - <<'EOF' || exit 1
import os, re, subprocess, sys, time, urllib.request, json
sys.path.insert(0, os.getcwd())
from k8s_gpu_monitor_amd._lib import binary
from k8s_gpu_monitor_amd.utils import promparse
env = dict(os.environ, GPUMON_PROF_PERIOD_S="10")
p = subprocess.Popen([binary("gpumon-exporter"), "-e", "--prof-counters=child", "--extras", "-o", "", "-d", "1000", "--self-metrics", "--listen", "127.0.0.1:0"],
                     stderr=subprocess.PIPE, text=True, env=env)
port = None
for _ in range(100):
    m = re.search(r"Serving /metrics on port (\d+)", p.stderr.readline() or "")
    if m:
        port = int(m.group(1)); break
assert port
def cpu(pid):
    f = open(f"/proc/{pid}/stat").read().rsplit(")", 1)[1].split()
    tck = os.sysconf("SC_CLK_TCK")
    return (int(f[11]) + int(f[12])) / tck, (int(f[13]) + int(f[14])) / tck   # self, reaped children
time.sleep(3)
s0, c0, t0 = *cpu(p.pid), time.time()
seen = []
for i in range(45):
    time.sleep(1)
    fam = promparse.parse(urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=5).read().decode())
    row = {k: (fam[k].samples[0][1] if k in fam and fam[k].samples else None)
           for k in ("dcgm_fi_prof_pipe_tensor_active", "dcgm_fi_prof_sm_active", "gpumon_prof_mem_pipe_active", "dcgm_fi_prof_gr_engine_active")}
    seen.append(row)
s1, c1, t1 = *cpu(p.pid), time.time()
rss = int(re.search(r"VmRSS:\s+(\d+)", open(f"/proc/{p.pid}/status").read()).group(1))
p.terminate(); p.wait(timeout=15)
res = {"seconds": round(t1 - t0, 1), "period_s": 10, "exporter_cpu_percent": round(100 * (s1 - s0) / (t1 - t0), 3),
       "probe_children_cpu_percent": round(100 * (c1 - c0) / (t1 - t0), 3), "exporter_rss_kb": rss,
       "scrapes_with_tensor_value": sum(1 for r in seen if r["dcgm_fi_prof_pipe_tensor_active"] is not None), "scrapes": len(seen),
       "last": seen[-1]}
print(json.dumps(res, indent=1))
open("gpurun_out/prof_grbm/exporter_child.json", "w").write(json.dumps(res, indent=1) + "\n")
EOF
echo "=== prof grbm complete"
