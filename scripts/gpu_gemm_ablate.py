"""Where does the pipelined GEMM spend its time?  Times the kernel with its fragment
reads, its LDS-DMA or its MFMAs removed (timing-only builds: make diag-ablate).
Writes gpurun_out/gemm_ablate.json."""
import json
import os
import sys

os.environ["GPUMON_DIAG_LIB"] = "libgpumon_diag_ablate.so"
import torch  # noqa: E402

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd import _lib  # noqa: E402

_orig_find = _lib._find
_lib._find = lambda name: _orig_find("libgpumon_diag_ablate.so" if name == "libgpumon_diag.so" else name)
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
NAMES = {1: "full", 25: "ahead 5", 26: "ahead 6", 27: "ahead 7", 11: "no reads", 12: "no dma", 13: "no reads, no dma", 14: "no mfma", 15: "no reads, no mfma",
         16: "no dma, no mfma", 17: "barriers only", 0: "simple loop",
         18: "full, operands L2-resident", 23: "dma only, operands L2-resident", 33: "dma only, ahead 5", 34: "dma only, ahead 6"}
if os.environ.get("GEMM_ABLATE_ONLY"):
    NAMES = {v: NAMES[v] for v in [1] + [int(x) for x in os.environ["GEMM_ABLATE_ONLY"].split(",")]}
out = []
# the issue-distance variants must compute the same bits as the simple loop, launch after launch
for dtype in ("bf16", "fp8"):
    for (m, n, k) in [(512, 768, 1024), (4096, 4096, 4096)]:
        a = torch.randn(m, k, device=dev)
        b = torch.randn(n, k, device=dev)
        a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
        c = torch.empty(m, n, dtype=torch.bfloat16, device=dev)
        diag.gemm_set_variant(0)
        diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream)
        torch.cuda.synchronize()
        ref = c.clone()
        for v in (25, 26, 27):
            diag.gemm_set_variant(v)
            bad = 0
            for i in range(60):
                diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), m, n, k, stream)
                if i % 10 == 9:
                    torch.cuda.synchronize()
                    bad += int(not torch.equal(c, ref))
            print("numerics", dtype, (m, n, k), "variant", v, "mismatching checks", bad, flush=True)
            out.append({"dtype": dtype, "shape": [m, n, k], "variant": v, "mismatches": bad})
for dtype in ("bf16", "fp8"):
    s = 8192
    a = torch.randn(s, s, device=dev)
    b = torch.randn(s, s, device=dev)
    a, b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
    c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
    best = {v: 1e9 for v in NAMES}
    for rep in range(3):
        for v in NAMES:
            diag.gemm_set_variant(v)
            diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                diag.gemm_launch(dtype, a.data_ptr(), b.data_ptr(), c.data_ptr(), s, s, s, stream)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) / 10)
    for v, name in NAMES.items():
        row = {"dtype": dtype, "variant": v, "what": name, "ms": round(best[v], 4), "rel_to_full": round(best[v] / best[1], 3)}
        out.append(row)
        print(row, flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_ablate.json", "w"), indent=1)
