"""Which schedule is fastest where the 256 x 256 tiles cannot fill the chip: large tiles (2), small tiles (8), small tiles
split along K (9), large tiles split along K (10) and the vendor's kernel, over M x N from 512^2 to 3072 x 2048 and K from
2048 to 16384.  Writes gpurun_out/gemm_split_sweep.json; the automatic choice (launch_gemm_t) is held against it."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream


def timed(fn, n=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n)
    return best


out = []
for dtype in ("bf16", "fp8"):
    for (m, n) in [(512, 512), (1024, 512), (1024, 1024), (1536, 1024), (2048, 1024), (1536, 1536), (2048, 2048), (3072, 2048), (4096, 2048)]:
        for k in (1024, 2048, 4096, 8192, 16384):
            a = torch.randn(m, k, device=dev)
            b = torch.randn(n, k, device=dev)
            a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
            c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
            row = {"dtype": dtype, "m": m, "n": n, "k": k}
            for variant in (-1, 2, 8, 9, 10):
                diag.gemm_set_variant(variant)
                us = timed(lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream)) * 1e3
                row["auto" if variant < 0 else f"v{variant}"] = round(us, 1)
                if variant < 0:
                    row["auto_schedule"], row["auto_splits"] = diag.gemm_plan(dtype, m, n, k)
                else:
                    row[f"s{variant}"] = diag.gemm_split_k(dtype, m, n, k)
            diag.gemm_set_variant(-1)
            bt = b.t()
            if dtype == "bf16":
                row["vendor"] = round(timed(lambda: torch.matmul(a, bt, out=c)) * 1e3, 1)
            else:
                one = torch.ones((), device=dev)
                row["vendor"] = round(timed(lambda: torch._scaled_mm(a, bt, scale_a=one, scale_b=one, out_dtype=torch.bfloat16)) * 1e3, 1)
            best = min(("v2", "v8", "v9", "v10"), key=lambda v: row[v])
            row["best"] = best
            print(row, flush=True)
            out.append(row)
            del a, b, c
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_split_sweep.json", "w"), indent=1)
