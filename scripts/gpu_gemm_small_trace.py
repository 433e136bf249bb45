#!/usr/bin/env python3
"""Back-to-back launches of the small-tile GEMM and of the vendor's kernel at shapes with few tiles, for a kernel
trace: kernel time against launch-to-launch time tells the K loop from the launch gap."""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from k8s_gpu_monitor_amd.ops import diag

dev = torch.device("cuda", 0)
s = torch.cuda.current_stream(dev).cuda_stream
for (m, n, k) in [(2048, 2048, 2048), (2048, 2048, 8192), (1024, 1024, 4096)]:
    a = torch.randn(m, k, device=dev, dtype=torch.bfloat16)
    b = torch.randn(n, k, device=dev, dtype=torch.bfloat16)
    c = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    for _ in range(3):
        diag.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, s)
        torch.matmul(a, b.t(), out=c)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    e0.record()
    for _ in range(200):
        diag.gemm_launch("bf16", a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, s)
    e1.record()
    for _ in range(200):
        torch.matmul(a, b.t(), out=c)
    e2.record()
    torch.cuda.synchronize()
    print(f"{m}x{n}x{k}: ours {e0.elapsed_time(e1) / 200 * 1e3:.2f} us per launch, vendor {e1.elapsed_time(e2) / 200 * 1e3:.2f} us", flush=True)
