"""Where a persistent workgroup's time goes between tiles (make diag-ablate, variants 53..55, GEMM_TL_VARIANTS):
per tile, real-time stamps at the top of the tile loop, when the first half-tiles (and the
previous tile's stores) have landed, and after the last MFMA.  Prints, per tile round, the
K-loop time, the wait at the top and the gap between the last MFMA of one tile and the top
of the next (prologue issue + store issue), and how far apart the CUs run.
Writes gpurun_out/gemm_timeline.json."""
import ctypes as C
import json
import os
import statistics
import sys

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
s = 8192
ntiles = (s // 256) ** 2
GRID = 256
for dtype, variant in [(d, v) for d in os.environ.get("GEMM_TL_DTYPES", "bf16,fp8").split(",")
                       for v in (int(x) for x in os.environ.get("GEMM_TL_VARIANTS", "53").split(","))]:
    a = torch.randn(s, s, device=dev)
    b = torch.randn(s, s, device=dev)
    a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
    c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
    diag.gemm_set_variant(variant)
    launch = lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 1500
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    buf = (C.c_ulonglong * (4 * ntiles))()
    assert lib.gmd_gemm_read_timeline(buf, ntiles) == 0
    tl = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(ntiles)]
    # tiles of one workgroup: chunk of 128 per XCD, local index j walks j, j+32, j+64, j+96
    per_xcd, step = ntiles // 8, GRID // 8
    t_first = min(t[0] for t in tl)
    t_last = max(t[2] for t in tl)
    rounds = per_xcd // step
    row = {"dtype": dtype, "variant": variant, "ms_per_launch": round(ms, 4), "kernel_span_us_top_to_last_mfma": (t_last - t_first) / 100.0, "rounds": []}
    for r in range(rounds):
        ids = [x * per_xcd + r * step + j for x in range(8) for j in range(step)]
        kloop = [(tl[i][2] - tl[i][1]) / 100.0 for i in ids]
        wait = [(tl[i][1] - tl[i][0]) / 100.0 for i in ids]
        cyc = [tl[i][3] for i in ids]
        rr = {"round": r, "kloop_us_median": statistics.median(kloop), "kloop_us_max": max(kloop), "kloop_cycles_median": statistics.median(cyc),
              "top_wait_us_median": statistics.median(wait), "top_wait_us_max": max(wait),
              "start_spread_us": (max(tl[i][0] for i in ids) - min(tl[i][0] for i in ids)) / 100.0,
              "end_spread_us": (max(tl[i][2] for i in ids) - min(tl[i][2] for i in ids)) / 100.0,
              "start_us_median": statistics.median((tl[i][0] - t_first) / 100.0 for i in ids)}
        if r + 1 < rounds:
            gap = [(tl[i + step][0] - tl[i][2]) / 100.0 for i in ids]
            rr["gap_to_next_top_us_median"] = statistics.median(gap)
        row["rounds"].append(rr)
        print(dtype, rr, flush=True)
    print(dtype, "ms/launch", row["ms_per_launch"], "span", row["kernel_span_us_top_to_last_mfma"], flush=True)
    out.append(row)
diag.gemm_set_variant(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_timeline.json", "w"), indent=1)
