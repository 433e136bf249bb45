"""MX-fp8 GEMM (e4m3 elements, one E8M0 scale per row and 32 elements of K) next to the vendor's block-scaled kernel
(hipBLASLt through torch._scaled_mm with float8_e8m0fnu scales) on the same operands in the same process: outputs compared
with each other and with fp32, back-to-back launches timed with device events.  Writes gpurun_out/gemm_mx_vendor.json.
If this torch build has no block-scaled path for the device the row says so and carries our figure alone."""
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


for size in (4096, 8192, 16384):
    m = n = k = size
    g = torch.Generator(device=dev).manual_seed(size)
    a = torch.randn(m, k, device=dev, generator=g).to(torch.float8_e4m3fn)
    b = torch.randn(n, k, device=dev, generator=g).to(torch.float8_e4m3fn)
    sa = torch.randint(124, 131, (m, k // 32), dtype=torch.uint8, device=dev, generator=g)
    sb = torch.randint(124, 131, (n, k // 32), dtype=torch.uint8, device=dev, generator=g)
    pa, pb = torch.empty_like(sa), torch.empty_like(sb)
    diag.mx_pack_scales(sa.data_ptr(), pa.data_ptr(), m, k, stream)
    diag.mx_pack_scales(sb.data_ptr(), pb.data_ptr(), n, k, stream)
    c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
    ms = timed(lambda: diag.gemm_launch("mxfp8", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream, scale_a=pa.data_ptr(),
                                        scale_b=pb.data_ptr()))
    row = {"m": m, "n": n, "k": k, "ms": round(ms, 4), "tflops": round(2.0 * m * n * k / ms / 1e9, 1)}
    try:
        ea, eb = sa.view(torch.float8_e8m0fnu), sb.view(torch.float8_e8m0fnu)
        bt = b.t()
        cv = torch._scaled_mm(a, bt, scale_a=ea, scale_b=eb, out_dtype=torch.bfloat16)
        torch.cuda.synchronize()
        diff = (cv.float() - c.float()).abs().max().item() / c.float().abs().max().item()
        tms = timed(lambda: torch._scaled_mm(a, bt, scale_a=ea, scale_b=eb, out_dtype=torch.bfloat16))
        row.update({"torch_ms": round(tms, 4), "torch_tflops": round(2.0 * m * n * k / tms / 1e9, 1), "max_rel_diff_vs_torch": diff})
    except Exception as e:
        row["torch_error"] = str(e).splitlines()[0][:200]
    print(row, flush=True)
    out.append(row)
    del a, b, c, sa, sb, pa, pb
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_mx_vendor.json", "w"), indent=1)
