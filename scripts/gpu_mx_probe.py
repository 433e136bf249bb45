"""Which C entries does one packed MX scale byte touch?  All-ones operands, unit scales except one byte."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
st = torch.cuda.current_stream().cuda_stream
m = n = 256
k = 128
one = torch.ones(m, k, device=dev).to(torch.float8_e4m3fn)
c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
diag.gemm_set_variant(0)
for which in ("A", "B"):
    for (unit, d, byte) in [(0, 0, 0), (0, 0, 1), (0, 1, 0), (0, 16, 0), (0, 37, 2), (1, 5, 3), (3, 63, 1)]:
        pa = torch.full((m // 64, 64), 0x7f7f7f7f, dtype=torch.int32, device=dev)
        pb = pa.clone()
        t = pa if which == "A" else pb
        v = 0x7f7f7f7f & ~(0xff << (8 * byte)) | (0x80 << (8 * byte))
        t[unit, d] = v - (1 << 32) if v >= (1 << 31) else v
        diag.gemm_launch("mxfp8", one.data_ptr(), one.data_ptr(), c.data_ptr(), m, n, k, st, scale_a=pa.data_ptr(), scale_b=pb.data_ptr())
        torch.cuda.synchronize()
        cf = c.float()
        base = cf.mode().values.mode().values.item()
        diff = (cf != base).nonzero()
        rows = sorted(set(diff[:, 0].tolist()))
        cols = sorted(set(diff[:, 1].tolist()))
        vals = sorted(set(cf[cf != base].tolist()))
        fq, fr = d >> 4, d & 15
        print(f"{which} unit {unit} dword {d} (fq {fq} fr {fr}) byte {byte}: base {base} changed {len(diff)} vals {vals[:4]} rows {rows[:6]}{'...' if len(rows) > 6 else ''} cols {cols[:6]}{'...' if len(cols) > 6 else ''}"
              f"  expect {'row' if which == 'A' else 'col'} {64 * unit + 16 * byte + fr}")

# which K block does the scale in dword fq*16+fr act on?  A is nonzero in one block only.
for q0 in range(4):
    a = torch.zeros(m, k, device=dev)
    a[:, 32 * q0:32 * q0 + 32] = 1
    a8 = a.to(torch.float8_e4m3fn)
    hits = []
    for fq in range(4):
        pa = torch.full((m // 64, 64), 0x7f7f7f7f, dtype=torch.int32, device=dev)
        pb = pa.clone()
        pa[0, fq * 16 + 0] = 0x7f7f7f80
        diag.gemm_launch("mxfp8", a8.data_ptr(), one.data_ptr(), c.data_ptr(), m, n, k, st, scale_a=pa.data_ptr(), scale_b=pb.data_ptr())
        torch.cuda.synchronize()
        hits.append(c[0, 0].item())
    print("A nonzero in block", q0, "-> C[0][0] with the scale of fq 0..3 doubled:", hits)

# the full path: random data, random scales, packed by gmd_mx_pack_scales
g = torch.Generator(device=dev).manual_seed(1)
a8 = torch.randn(m, k, device=dev, generator=g).to(torch.float8_e4m3fn)
b8 = torch.randn(n, k, device=dev, generator=g).to(torch.float8_e4m3fn)
sa = torch.randint(124, 131, (m, k // 32), dtype=torch.uint8, device=dev, generator=g)
sb = torch.randint(124, 131, (n, k // 32), dtype=torch.uint8, device=dev, generator=g)
pa, pb = torch.empty_like(sa), torch.empty_like(sb)
diag.mx_pack_scales(sa.data_ptr(), pa.data_ptr(), m, k, st)
diag.mx_pack_scales(sb.data_ptr(), pb.data_ptr(), n, k, st)
torch.cuda.synchronize()
# packed layout computed in torch: unit [rg][kt], dword fq*16+fr, byte mb
exp = sa.view(m // 64, 4, 16, k // 128, 4).permute(0, 3, 4, 2, 1).contiguous().view(-1)  # [rg][kt][fq][fr][mb]
print("pack kernel matches the layout:", bool(torch.equal(exp, pa.view(-1))))
diag.gemm_launch("mxfp8", a8.data_ptr(), b8.data_ptr(), c.data_ptr(), m, n, k, st, scale_a=pa.data_ptr(), scale_b=pb.data_ptr())
torch.cuda.synchronize()
dq = lambda x, s: x.float() * torch.exp2(s.float() - 127).repeat_interleave(32, dim=1)
ref = dq(a8, sa) @ dq(b8, sb).t()
print("err", ((c.float() - ref).abs().max() / ref.abs().max()).item())
