"""Cycles or clock?  For each ablation arm of the pipelined GEMM (make diag-ablate)
run back-to-back launches for a while, then read the in-kernel stamps of the last
launch: shader-clock ticks per workgroup (the cycles the schedule costs) and the
clock the chip held (ticks / 100 MHz real-time ticks).  Wall time per launch comes
from events around the same run.  Writes gpurun_out/gemm_clock.json."""
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
lib.gmd_gemm_read_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
SOAK_S = float(os.environ.get("SOAK_S", "2.0"))
ARMS = {41: "full", 42: "mfma + barriers", 43: "no reads", 44: "no dma", 45: "dma only", 46: "full, operands L2-resident",
        51: "four waves x 128x128", 62: "four waves: mfma + barriers", 63: "four waves: no reads", 64: "four waves: no dma",
        65: "four waves: dma only", 66: "four waves: no barriers", 67: "four waves: mfma alone", 68: "four waves: pieces back to back", 69: "four waves: reads 2 apart, pieces 4 apart",
        70: "four waves: pieces 5 apart", 71: "four waves: reads 2 apart, pieces 3 apart"}
if os.environ.get("GEMM_CLOCK_ONLY"):
    ARMS = {int(v): ARMS[int(v)] for v in os.environ["GEMM_CLOCK_ONLY"].split(",")}
out = []
for dtype in ("bf16", "fp8"):
    s = 8192
    a = torch.randn(s, s, device=dev)
    b = torch.randn(s, s, device=dev)
    a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
    c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
    ntiles = (s // 256) ** 2
    for v, name in ARMS.items():
        diag.gemm_set_variant(v)
        launch = lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
        launch()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        torch.cuda.synchronize()
        n = max(20, int(SOAK_S * 1e3 / max(e0.elapsed_time(e1), 0.05)))
        e0.record()
        for _ in range(n):
            launch()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        buf = (C.c_ulonglong * (2 * ntiles))()
        rc = lib.gmd_gemm_read_stamps(buf, ntiles)
        assert rc == 0, rc
        ticks = [buf[2 * i] for i in range(ntiles)]
        real = [buf[2 * i + 1] for i in range(ntiles)]
        mhz = [100.0 * t / r for t, r in zip(ticks, real) if r]
        row = {"dtype": dtype, "variant": v, "what": name, "ms_per_launch": round(ms, 4), "launches": n,
               "wg_cycles_median": int(statistics.median(ticks)), "clock_mhz_median": round(statistics.median(mhz), 1),
               "wg_us_median": round(statistics.median(real) / 100.0, 2)}
        out.append(row)
        print(row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_clock.json", "w"), indent=1)
