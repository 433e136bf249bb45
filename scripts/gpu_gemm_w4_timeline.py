"""Where one tile's time goes in the four-wave GEMM (make diag-ablate, variant 51): per workgroup, shader-clock
ticks from the kernel's first instruction to (1) prologue done (first fragments in registers), (2) last MFMA
issued, (3) last LDS-DMA landed, (4) last store issued.  Prints medians per dtype; writes
gpurun_out/gemm_w4_timeline.json."""
import ctypes as C
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd import _lib  # noqa: E402

_orig_find = _lib._find
_lib._find = lambda name: _orig_find(os.environ.get("GEMM_TL_LIB", "libgpumon_diag_ablate.so") if name == "libgpumon_diag.so" else name)
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

lib = diag._lib()
lib.gmd_gemm_read_timeline.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
s = int(os.environ.get("GEMM_TL_SIZE", "8192"))
ntiles = min((s // 256) ** 2, 4096)
out = []
for dtype in os.environ.get("GEMM_TL_DTYPES", "bf16,fp8").split(","):
    a = torch.randn(s, s, device=dev)
    b = torch.randn(s, s, device=dev)
    a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
    c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
    kw = {}
    if dtype == "mxfp8":
        sc = torch.randint(124, 131, (s, s // 32), dtype=torch.uint8, device=dev)
        pk = torch.empty_like(sc)
        diag.mx_pack_scales(sc.data_ptr(), pk.data_ptr(), s, s, stream)
        kw = {"scale_a": pk.data_ptr(), "scale_b": pk.data_ptr()}
    diag.gemm_set_variant(int(os.environ.get("GEMM_TL_VARIANT", "51")))
    launch = lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream, **kw)
    for _ in range(5):
        launch()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 500 if s <= 8192 else 60
    for _ in range(n):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    buf = (C.c_ulonglong * (4 * ntiles))()
    assert lib.gmd_gemm_read_timeline(buf, ntiles) == 0
    tl = [(buf[4 * i], buf[4 * i + 1], buf[4 * i + 2], buf[4 * i + 3]) for i in range(ntiles)]
    med = lambda xs: statistics.median(xs)
    if os.environ.get("GEMM_TL_VARIANT") == "78":  # per wave: ticks to the end of the prologue's wait for data
        row = {"dtype": dtype, "size": s, "wave_ticks_median": [med(t[w] for t in tl) for w in range(4)],
               "spread_median": med(max(t) - min(t) for t in tl)}
        print(row, flush=True)
        out.append(row)
        continue
    if os.environ.get("GEMM_TL_VARIANT") == "76":  # the prologue in four steps (ticks from the first instruction)
        row = {"dtype": dtype, "size": s, "ms_per_launch": round(ms, 4), "before_first_dma": med(t[0] for t in tl),
               "dma_issued": med(t[1] for t in tl), "first_half_tiles_landed": med(t[2] for t in tl), "fragments_read": med(t[3] for t in tl)}
        print(row, flush=True)
        out.append(row)
        continue
    row = {"dtype": dtype, "size": s, "ms_per_launch": round(ms, 4),
           "prologue_ticks": med(t[0] for t in tl), "k_loop_ticks": med(t[1] - t[0] for t in tl),
           "last_dma_wait_ticks": med(t[2] - t[1] for t in tl), "epilogue_ticks": med(t[3] - t[2] for t in tl),
           "lifetime_ticks": med(t[3] for t in tl),
           "p90": {"prologue": sorted(t[0] for t in tl)[ntiles * 9 // 10], "k_loop": sorted(t[1] - t[0] for t in tl)[ntiles * 9 // 10],
                   "last_dma_wait": sorted(t[2] - t[1] for t in tl)[ntiles * 9 // 10], "epilogue": sorted(t[3] - t[2] for t in tl)[ntiles * 9 // 10]}}
    print(row, flush=True)
    out.append(row)
diag.gemm_set_variant(-1)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_w4_timeline.json", "w"), indent=1)
