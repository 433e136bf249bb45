"""GEMM kernel numerics vs torch (oracle) + throughput on a real MI355X."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_gpu_monitor_amd.ops import diag as D
out = {}
dev = torch.device("cuda:0")
torch.manual_seed(0)
def check_bf16(m, n, k):
    # A = I-check with asymmetric B first, then random
    a = torch.randn(m, k, device=dev).to(torch.bfloat16)
    b = torch.randn(n, k, device=dev).to(torch.bfloat16)
    c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    D.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k)
    torch.cuda.synchronize()
    ref = a.float() @ b.float().t()
    err = (c.float() - ref).abs().max().item()
    rel = err / ref.abs().max().item()
    return {"max_abs_err": err, "rel_to_max": rel, "ref_max": ref.abs().max().item()}
for shape in [(256, 256, 64), (256, 256, 128), (512, 768, 256), (1024, 1024, 4096)]:
    out[f"bf16_{shape}"] = check_bf16(*shape)
    print(shape, out[f"bf16_{shape}"], flush=True)
# identity check with asymmetric B: C = I_k @ B^T => C[m][n] = B[n][m]
m = n = k = 256
a = torch.eye(m, device=dev, dtype=torch.bfloat16)
b = (torch.arange(n * k, device=dev).reshape(n, k) % 251).to(torch.bfloat16)
c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
D.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k)
torch.cuda.synchronize()
out["identity_ok"] = bool(torch.equal(c, b.t().contiguous()))
print("identity", out["identity_ok"], flush=True)
# fp8 (MX-scaled, unit scales) vs torch float8_e4m3fn oracle
def check_fp8(m, n, k):
    a8 = (torch.randn(m, k, device=dev) * 0.5).to(torch.float8_e4m3fn)
    b8 = (torch.randn(n, k, device=dev) * 0.5).to(torch.float8_e4m3fn)
    c = torch.zeros(m, n, device=dev, dtype=torch.bfloat16)
    D.gemm_launch("mxfp8", a8.data_ptr(), b8.data_ptr(), c.data_ptr(), m, n, k)
    torch.cuda.synchronize()
    ref = a8.float() @ b8.float().t()
    err = (c.float() - ref).abs().max().item()
    return {"max_abs_err": err, "rel_to_max": err / ref.abs().max().item()}
try:
    for shape in [(256, 256, 128), (512, 512, 1024)]:
        out[f"fp8_{shape}"] = check_fp8(*shape)
        print("fp8", shape, out[f"fp8_{shape}"], flush=True)
except Exception as e:
    out["fp8_error"] = repr(e); print(out["fp8_error"], flush=True)
# throughput
for dt, shape in [("bf16", (4096, 4096, 4096)), ("bf16", (8192, 8192, 8192)), ("mxfp8", (8192, 8192, 8192))]:
    try:
        out[f"stress_{dt}_{shape[0]}"] = D.gemm_stress(dt, *shape, seconds=2.0, verify=True)
    except Exception as e:
        out[f"stress_{dt}_{shape[0]}"] = repr(e)
    print("stress", dt, shape, out[f"stress_{dt}_{shape[0]}"], flush=True)
# torch (hipBLASLt) reference rate on the same shape/data class
a = torch.randn(8192, 8192, device=dev).to(torch.bfloat16); b = torch.randn(8192, 8192, device=dev).to(torch.bfloat16)
for _ in range(3): (a @ b.t())
torch.cuda.synchronize(); t = time.time()
for _ in range(20): (a @ b.t())
torch.cuda.synchronize(); dt_ = (time.time() - t) / 20
out["torch_bf16_8192_tflops"] = 2 * 8192**3 / dt_ / 1e12
print("torch", out["torch_bf16_8192_tflops"], flush=True)
json.dump(out, open("gpurun_out/gemm_check.json", "w"), indent=1, default=str)
