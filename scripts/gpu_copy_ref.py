"""How the stream kernels compare with what the vendor stack does for the same
bytes on the same device, same process, interleaved rounds: hipMemcpyAsync D2D
(torch Tensor.copy_), torch's elementwise kernels (mul, add, addcmul-free triad)
and torch.sum for the read stream.  Writes gpurun_out/copy_ref.json."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from k8s_gpu_monitor_amd.ops import diag as D

dev = torch.device("cuda", 0)
N = (4 << 30) // 4  # 4 GiB per buffer
a = torch.ones(N, device=dev)
b = torch.full((N,), 2.0, device=dev)
c = torch.full((N,), 0.5, device=dev)
st = torch.cuda.current_stream().cuda_stream


def timed(fn, rounds=7):
    best = 1e9
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


GB = N * 4 / 1e12
sink = torch.zeros(32768, device=dev)  # GMD_STREAM_SINK_FLOATS
pa, pb, pc, ps = a.data_ptr(), b.data_ptr(), c.data_ptr(), sink.data_ptr()
L = D.hbm_stream_launch
cases = {
    "copy": (2, lambda: c.copy_(a), lambda: L("copy", pa, 0, pc, 3.0, N, "auto", st)),
    "scale": (2, lambda: torch.mul(c, 3.0, out=b), lambda: L("scale", pa, pb, pc, 3.0, N, "auto", st)),
    "add": (3, lambda: torch.add(a, b, out=c), lambda: L("add", pa, pb, pc, 3.0, N, "auto", st)),
    "read": (1, lambda: a.sum(), lambda: L("read", pa, 0, ps, 3.0, N, "auto", st)),
    "write": (1, lambda: a.fill_(3.0), lambda: L("write", pa, 0, 0, 3.0, N, "auto", st)),
}
out = {}
for name, (factor, vendor, ours) in cases.items():
    vendor(); ours()
    torch.cuda.synchronize()
    tv, to = [], []
    for _ in range(3):  # interleave so both see the same clocks and temperature
        tv.append(timed(vendor, 5))
        to.append(timed(ours, 5))
    out[name] = {"vendor_tbps": round(factor * GB / (min(tv) * 1e-3), 3), "gpumon_tbps": round(factor * GB / (min(to) * 1e-3), 3)}
    out[name]["ratio"] = round(out[name]["gpumon_tbps"] / out[name]["vendor_tbps"], 3)
    print(name, out[name], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/copy_ref.json", "w"), indent=1)
