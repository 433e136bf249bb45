"""When and where the workgroups of the four-wave GEMM ran (make diag-ablate, variant 75): per workgroup the
100 MHz real-time at its first instruction and behind its last store, and the CU it ran on (HW_ID, XCC_ID).
Prints how many workgroups each CU ran, the gap between one workgroup's last store and the first instruction
of the next on the same CU, how far apart the CUs start and finish, and the share of the kernel's span in
which a CU had no workgroup.  Writes gpurun_out/gemm_w4_where.json."""
import ctypes as C
import json
import os
import statistics
import sys
from collections import defaultdict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd import _lib  # noqa: E402

_orig_find = _lib._find
_lib._find = lambda name: _orig_find("libgpumon_diag_ablate.so" if name == "libgpumon_diag.so" else name)
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

lib = diag._lib()
lib.gmd_gemm_read_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = []
for s in (int(x) for x in os.environ.get("GEMM_WHERE_SIZES", "8192").split(",")):
    ntiles = (s // 256) ** 2
    a = torch.randn(s, s, device=dev).bfloat16()
    b = torch.randn(s, s, device=dev).bfloat16()
    c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
    diag.gemm_set_variant(75)
    for _ in range(200):
        diag.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
    torch.cuda.synchronize()
    n = min(ntiles, 4096)
    buf = (C.c_ulonglong * (4 * n))()
    assert lib.gmd_gemm_read_timeline(buf, n) == 0
    per_cu = defaultdict(list)
    for i in range(n):
        start, end, where = buf[4 * i], buf[4 * i + 1], buf[4 * i + 2]
        key = (where >> 32, (where >> 8) & 0xFF)  # XCC, (SE, SH, CU)
        per_cu[key].append((start, end))
    t0 = min(x[0] for v in per_cu.values() for x in v)
    t1 = max(x[1] for v in per_cu.values() for x in v)
    gaps, firsts, lasts, busy = [], [], [], []
    for v in per_cu.values():
        v.sort()
        firsts.append(v[0][0] - t0)
        lasts.append(t1 - v[-1][1])
        busy.append(sum(e - b_ for b_, e in v))
        gaps += [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    us = lambda ticks: round(ticks / 100.0, 2)
    row = {"size": s, "cus_seen": len(per_cu), "workgroups_per_cu": sorted(set(len(v) for v in per_cu.values())),
           "span_us": us(t1 - t0), "workgroup_us_median": us(statistics.median(e - b_ for v in per_cu.values() for b_, e in v)),
           "turnaround_us": {"median": us(statistics.median(gaps)), "p90": us(sorted(gaps)[len(gaps) * 9 // 10]), "max": us(max(gaps))} if gaps else None,
           "first_start_us": {"median": us(statistics.median(firsts)), "max": us(max(firsts))},
           "idle_at_end_us": {"median": us(statistics.median(lasts)), "max": us(max(lasts))},
           "cu_busy_share_median": round(statistics.median(busy) / (t1 - t0), 4)}
    per_xcc = defaultdict(list)
    for (xcc, _), v in per_cu.items():
        per_xcc[xcc].append(t1 - v[-1][1])
    row["idle_at_end_us_by_xcc"] = {int(k): us(statistics.median(v)) for k, v in sorted(per_xcc.items())}
    print(row, flush=True)
    out.append(row)
diag.gemm_set_variant(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_w4_where.json", "w"), indent=1)
