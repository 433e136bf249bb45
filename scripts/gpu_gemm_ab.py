"""A/B of the GEMM schedules on random data: numerics against fp32 matmul, then
interleaved timing (the variants alternate inside one process) at several shapes,
next to torch.matmul (hipBLASLt) for bf16.  Writes gpurun_out/gemm_ab.json.
GEMM_AB_VARIANTS (default 0,1,2,3,4,5), GEMM_AB_SIZES (default 4096,8192,16384) and GEMM_AB_PERSISTENT
(variants that run as a persistent grid, default 3) choose what runs."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if os.environ.get("GEMM_AB_ABLATE"):  # the ablation build (make diag-ablate) carries the extra arms (variants > 6)
    from k8s_gpu_monitor_amd import _lib

    _orig_find = _lib._find
    _lib._find = lambda name: _orig_find("libgpumon_diag_ablate.so" if name == "libgpumon_diag.so" else name)
from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream().cuda_stream
out = {"numerics": [], "timing": []}
VARIANTS = tuple(int(x) for x in os.environ.get("GEMM_AB_VARIANTS", "0,1,2,3,4,5").split(","))
SIZES = tuple(int(x) for x in os.environ.get("GEMM_AB_SIZES", "4096,8192,16384").split(","))
PERSISTENT = tuple(int(x) for x in os.environ.get("GEMM_AB_PERSISTENT", "3").split(",") if x)


class Case:
    """Operands of one GEMM; MX-fp8 carries random E8M0 scales (2^-3 .. 2^3), row-major for the reference and packed."""

    def __init__(self, m, n, k, dtype, seed=0):
        g = torch.Generator(device=dev).manual_seed(seed)
        a = torch.randn(m, k, device=dev, generator=g)
        b = torch.randn(n, k, device=dev, generator=g)
        self.m, self.n, self.k, self.dtype = m, n, k, dtype
        self.a, self.b = (a.bfloat16(), b.bfloat16()) if dtype == "bf16" else (a.to(torch.float8_e4m3fn), b.to(torch.float8_e4m3fn))
        self.kw, self.sa, self.sb = {}, None, None
        if dtype == "mxfp8":
            self.sa = torch.randint(124, 131, (m, k // 32), dtype=torch.uint8, device=dev, generator=g)
            self.sb = torch.randint(124, 131, (n, k // 32), dtype=torch.uint8, device=dev, generator=g)
            self.pa, self.pb = torch.empty_like(self.sa), torch.empty_like(self.sb)
            diag.mx_pack_scales(self.sa.data_ptr(), self.pa.data_ptr(), m, k, stream)
            diag.mx_pack_scales(self.sb.data_ptr(), self.pb.data_ptr(), n, k, stream)
            self.kw = {"scale_a": self.pa.data_ptr(), "scale_b": self.pb.data_ptr()}

    def ref(self):
        dq = lambda x, s: x.float() if s is None else x.float() * torch.exp2(s.float() - 127).repeat_interleave(32, dim=1)
        return dq(self.a, self.sa) @ dq(self.b, self.sb).t()

    def launch(self, c):
        diag.gemm_launch(self.dtype, self.a.data_ptr(), self.b.data_ptr(), c.data_ptr(), self.m, self.n, self.k, stream, **self.kw)


DTYPES = tuple(os.environ.get("GEMM_AB_DTYPES", "bf16,fp8,mxfp8").split(","))
ok_all = True
for dtype in DTYPES:
    for (m, n, k) in [(256, 256, 256), (512, 768, 1024), (1024, 256, 4096), (2048, 2048, 2048), (256, 512, 64 * 6 if dtype == "bf16" else 128 * 3),
                      (4352, 4352, 512), (8192, 4096, 256), (2304, 8192, 1280)]:  # > 256 tiles: the persistent grid loops
        case = Case(m, n, k, dtype, seed=m + n + k)
        ref = case.ref()
        res = {}
        for v in VARIANTS:
            diag.gemm_set_variant(v)
            c = torch.full((m, n), float("nan"), dtype=torch.bfloat16, device=dev)
            for _ in range(3):  # repeated: a race would show as run-to-run differences
                case.launch(c)
            torch.cuda.synchronize()
            res[v] = c.clone()
            err = ((c.float() - ref).abs().max() / ref.abs().max()).item()
            ok = bool(torch.isfinite(c.float()).all()) and err < 6e-3
            ok_all &= ok
            out["numerics"].append({"dtype": dtype, "shape": [m, n, k], "variant": v, "rel_err": err, "ok": ok})
        same = all(bool(torch.equal(res[VARIANTS[0]], res[v])) for v in VARIANTS)
        ok_all &= same
        out["numerics"].append({"dtype": dtype, "shape": [m, n, k], "variants_bit_identical": same})
        print(dtype, (m, n, k), "err", [round(x["rel_err"], 5) for x in out["numerics"][-1 - len(VARIANTS):-1]], "identical", same,
              flush=True)

# race screen: many launches of the pipeline variant must give one bit pattern
for rv, dtype in [(v, d) for v in VARIANTS if v != 0 for d in DTYPES]:
    diag.gemm_set_variant(rv)
    S = 8192 if rv in PERSISTENT else 4096  # the persistent grid needs more tiles than CUs
    case = Case(S, S, 4096, dtype, seed=5)
    c = torch.empty(S, S, dtype=torch.bfloat16, device=dev)
    case.launch(c)
    torch.cuda.synchronize()
    first = c.clone()
    bad = 0
    for i in range(200):
        case.launch(c)
        if i % 20 == 19:
            torch.cuda.synchronize()
            bad += int(not torch.equal(c, first))
    ref = case.ref()
    err = ((first.float() - ref).abs().max() / ref.abs().max()).item()
    out["numerics"].append({"dtype": dtype, "variant": rv, "race_screen_mismatches": bad, "rel_err_4096": err})
    ok_all &= bad == 0 and err < 6e-3
    print(dtype, "variant", rv, "race screen mismatches", bad, "err", err, flush=True)


def timeit(fn, n):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for dtype in DTYPES:
    for s in SIZES:
        case = Case(s, s, s, dtype, seed=s)
        c = torch.empty(s, s, dtype=torch.bfloat16, device=dev)
        flops = 2.0 * s ** 3
        best = {v: 1e9 for v in VARIANTS + ("torch",)}
        allms = {v: [] for v in VARIANTS + ("torch",)}
        n = 20 if s <= 8192 else 5
        for rep in range(int(os.environ.get("GEMM_AB_REPS", "4"))):  # interleaved A/B
            for v in VARIANTS:
                diag.gemm_set_variant(v)
                ms = timeit(lambda: case.launch(c), n)
                best[v] = min(best[v], ms)
                allms[v].append(ms)
            if dtype == "bf16":
                bt = case.b.t()
                ms = timeit(lambda: torch.matmul(case.a, bt), n)
                best["torch"] = min(best["torch"], ms)
                allms["torch"].append(ms)
        row = {"dtype": dtype, "size": s}
        row.update({f"tflops_v{v}": flops / best[v] / 1e9 for v in VARIANTS})
        row.update({f"median_ms_v{v}": sorted(allms[v])[len(allms[v]) // 2] for v in VARIANTS})
        if dtype == "bf16":
            row["tflops_torch"] = flops / best["torch"] / 1e9
            row["median_ms_torch"] = sorted(allms["torch"])[len(allms["torch"]) // 2]
        out["timing"].append(row)
        print(row, flush=True)
        del case, c
diag.gemm_set_variant(-1)
out["ok"] = ok_all
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/gemm_ab.json", "w"), indent=1)
print("OK" if ok_all else "NUMERICS FAILED")
sys.exit(0 if ok_all else 1)
