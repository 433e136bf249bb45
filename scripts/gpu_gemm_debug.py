"""Localise a wrong GEMM variant: which 16-byte K chunks of A meet which chunks of B,
and which rows / columns of C come out wrong.  usage: gpu_gemm_debug.py VARIANT DTYPE"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

variant, dtype = int(sys.argv[1]), sys.argv[2]
dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
el = 8 if dtype == "bf16" else 16  # elements per 16-byte chunk
kt_el = 8 * el                      # elements per 128-byte K-tile
m = n = 256
NKT = int(sys.argv[3]) if len(sys.argv) > 3 else 2
k = NKT * kt_el
tdt = torch.bfloat16 if dtype == "bf16" else torch.float8_e4m3fn


def run(a, b, v):
    diag.gemm_set_variant(v)
    c = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=dev)
    ah, bh = a.to(tdt), b.to(tdt)  # keep the converted operands alive across the launch
    diag.gemm_launch(dtype, ah.data_ptr(), bh.data_ptr(), c.data_ptr(), m, n, k, stream)
    torch.cuda.synchronize()
    return c.float()


# 1. chunk pairing
print("chunk pairing (rows: A chunk, cols: B chunk; expected: identity)")
for ka in range(8):
    row = []
    for kb in range(8):
        a = torch.zeros(m, k, device=dev)
        b = torch.zeros(n, k, device=dev)
        for t in range(NKT):
            a[:, t * kt_el + ka * el:t * kt_el + (ka + 1) * el] = 1.0
            b[:, t * kt_el + kb * el:t * kt_el + (kb + 1) * el] = 1.0
        ah, bh = a.to(tdt), b.to(tdt)
        diag.gemm_set_variant(variant)
        c = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=dev)
        diag.gemm_launch(dtype, ah.data_ptr(), bh.data_ptr(), c.data_ptr(), m, n, k, stream)
        torch.cuda.synchronize()
        row.append(round(c.float().mean().item(), 1))
    print(ka, row, flush=True)

# 2. row / column map: A row i = i+1 in chunk 0 only, B all ones -> C[i][j] = el*2*(i+1)
a = torch.zeros(m, k, device=dev)
b = torch.ones(n, k, device=dev)
a[:, 0] = torch.arange(1, m + 1, device=dev).float() / 4
c = run(a, b, variant)
want = (a.to(tdt).float() @ b.to(tdt).float().t())
bad_rows = (c != want).any(dim=1).nonzero().flatten().tolist()
print("rows wrong with A-row ramp:", bad_rows[:40], len(bad_rows))
# got*4 = the (1-based) A row whose value arrived; per 16-row block, columns 0 and 16
for mb in range(8):
    print("  mb", mb, "col 0 :", [int(x) for x in (c[mb * 16:(mb + 1) * 16, 0] * 4 / 1).tolist()])
    print("  mb", mb, "col 16:", [int(x) for x in (c[mb * 16:(mb + 1) * 16, 16] * 4 / 1).tolist()])
a = torch.ones(m, k, device=dev)
b = torch.zeros(n, k, device=dev)
b[:, 0] = torch.arange(1, n + 1, device=dev).float() / 4
c = run(a, b, variant)
want = (a.to(tdt).float() @ b.to(tdt).float().t())
bad_cols = (c != want).any(dim=0).nonzero().flatten().tolist()
print("cols wrong with B-row ramp:", bad_cols[:40], len(bad_cols))
# 2b. random data: which 16x16 blocks of C are wrong (rows of the printout = 16-row blocks)
torch.manual_seed(1)
a = torch.randn(m, k, device=dev)
b = torch.randn(n, k, device=dev)
c = run(a, b, variant)
want = a.to(tdt).float() @ b.to(tdt).float().t()
err = ((c - want).abs() > 0.05 * want.abs().max()).float().view(16, 16, 16, 16).amax(dim=(1, 3))
print("wrong 16x16 blocks (1 = wrong):")
for r in range(16):
    print("".join("1" if x else "." for x in err[r].tolist()))
# 3. which K-tile: ones in K-tile t only
for t in range(NKT):
    a = torch.zeros(m, k, device=dev)
    b = torch.ones(n, k, device=dev)
    a[:, t * kt_el:(t + 1) * kt_el] = 1.0
    c = run(a, b, variant)
    print("K-tile", t, "only: C values", sorted(set(c.flatten().tolist()))[:8], "expected", float(kt_el))
