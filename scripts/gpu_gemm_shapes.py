"""What the general instantiation of the four-wave GEMM costs: shapes with partial tiles and odd K-tile counts next to
the nearest full-tile, even-count shape, bf16 and fp8, with the vendor's kernel on the same operands (hipBLASLt through
torch.matmul for bf16, through torch._scaled_mm with unit scales for fp8).  Writes gpurun_out/gemm_shapes.json."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = []


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


for dtype in ("bf16", "fp8"):
    kt = 64 if dtype == "bf16" else 128
    for (m, n, k) in [(512, 512, 8192), (1024, 1024, 2048), (1024, 1024, 4096), (1024, 1024, 16384), (1024, 2048, 8192), (2048, 2048, 2048), (2048, 2048, 8192), (3072, 3072, 3072), (2048, 4096, 4096), (4096, 4096, 4096), (8192, 8192, 8192), (16384, 16384, 16384), (8192, 8192, 8192 + kt), (8200, 8200, 8192), (8000, 8000, 8192), (4000, 4104, 4096 + kt),
                      (1000, 520, 5 * kt)]:
        a = torch.randn(m, k, device=dev)
        b = torch.randn(n, k, device=dev)
        a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        ms = timed(lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream))
        row = {"dtype": dtype, "m": m, "n": n, "k": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
        if diag.gemm_split_k(dtype, m, n, k) > 1:
            row["split_k"] = diag.gemm_split_k(dtype, m, n, k)
        # the two tile sizes the default chooses between: 256 x 256 (schedule 2) and 128 x 128 (schedule 8)
        for name, variant in (("large", 2), ("small", 8)):
            if variant == 8 and m * n > 8192 * 8192:
                continue
            diag.gemm_set_variant(variant)
            row[f"ms_{name}_tiles"] = round(timed(lambda: diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream)), 4)
        diag.gemm_set_variant(-1)
        if dtype == "bf16":
            bt = b.t()
            tms = timed(lambda: torch.matmul(a, bt, out=c))
            row.update({"torch_ms": round(tms, 4), "torch_tflops": round(2.0 * m * n * k / tms / 1e9, 1)})
        else:
            one = torch.ones((), device=dev)
            bt = b.t()
            try:
                tms = timed(lambda: torch._scaled_mm(a, bt, scale_a=one, scale_b=one, out_dtype=torch.bfloat16))
                row.update({"torch_ms": round(tms, 4), "torch_tflops": round(2.0 * m * n * k / tms / 1e9, 1)})
            except Exception as e:  # shapes the vendor's fp8 path does not take (it wants multiples of 16)
                row["torch_error"] = str(e).splitlines()[0][:120]
        del a, b, c
        print(row, flush=True)
        out.append(row)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_shapes.json", "w"), indent=1)
