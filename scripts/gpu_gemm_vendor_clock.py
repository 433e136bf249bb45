"""Launch list for a rocprofv3 counter pass that compares the clock the chip holds under this
repository's bf16 GEMM and under hipBLASLt's (torch.matmul), same operands, same process:
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d OUT -- python3 scripts/gpu_gemm_vendor_clock.py
Effective clock of a dispatch = GRBM_GUI_ACTIVE / 8 / duration (the counter sums the 8 XCDs).
GEMM_VC_REPS (default 3) and GEMM_VC_LAUNCHES (default 150) shorten it for other counter passes
(instruction mix, L2 traffic: scripts/gpu_gemm_vendor_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
s = int(os.environ.get("GEMM_VC_SIZE", "8192"))
a = torch.randn(s, s, device=dev).bfloat16()
b = torch.randn(s, s, device=dev).bfloat16()
c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
bt = b.t()
REPS = int(os.environ.get("GEMM_VC_REPS", "3"))
LAUNCHES = int(os.environ.get("GEMM_VC_LAUNCHES", "150"))
for rep in range(REPS):  # interleaved blocks of launches, long enough for the clock to settle
    for v in (3, 1, 2):
        diag.gemm_set_variant(v)
        for _ in range(LAUNCHES):
            diag.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
    for _ in range(LAUNCHES):
        torch.matmul(a, bt, out=c)
    torch.cuda.synchronize()
print("done")
