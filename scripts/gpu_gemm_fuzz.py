#!/usr/bin/env python3
"""Random shapes through the GEMM's automatic schedule, each held against the fp32 product of the same operands.

What the fixed shapes of tests/test_gpu_diag.py cannot reach by enumeration: every pairing of ragged M / N (multiples
of 8), odd and long K-tile counts, and the planner's choices between them (large tiles, small tiles, strips, a split
along K on either tile size with a short last range).  Per case: relative error against fp32 under 6e-3, a guard band
behind C untouched, and the same bits from a second launch.  One JSON line per case in gpurun_out/gemm_fuzz.jsonl; the
exit status is 1 when a case fails.

With --soak SECONDS the same shapes go through the soak driver instead (operands of its own, its fp32 reference check,
every launch summed in the kernel and every result re-read from memory): the summing forms of the kernels are other
instantiations than the ones a plain launch runs, and a sum that came out differently from launch to launch at some
shape would be reported to an operator as silent data corruption.

    python scripts/gpu_gemm_fuzz.py [--cases 150] [--seed 1] [--soak 0.05]
"""
import argparse
import json
import os
import random
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=150)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/gemm_fuzz.jsonl")
    ap.add_argument("--soak", type=float, default=0.0, help="seconds per case through the soak driver instead of single launches")
    a = ap.parse_args()
    import torch
    from k8s_gpu_monitor_amd.ops import diag
    dev = torch.device("cuda", 0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    rnd = random.Random(a.seed)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    failed, by_schedule, t_last = 0, {}, time.time()

    def shape(dtype):
        kind = rnd.choice(["few_tiles_long_k", "ragged", "tiny", "mid", "large_long_k"])
        step = 256 if dtype == "mxfp8" else 8
        if kind == "few_tiles_long_k":
            m, n, kt = rnd.randrange(step, 2305, step), rnd.randrange(step, 2305, step), rnd.randrange(16, 261)
        elif kind == "ragged":
            m, n, kt = rnd.randrange(step, 4401, step), rnd.randrange(step, 4401, step), rnd.randrange(1, 40)
        elif kind == "large_long_k":
            m, n, kt = rnd.randrange(1536, 4097, step), rnd.randrange(1536, 4097, step), rnd.randrange(64, 261)
        elif kind == "tiny":
            m, n, kt = rnd.randrange(step, 520, step), rnd.randrange(step, 520, step), rnd.randrange(1, 300)
        else:
            m, n, kt = rnd.randrange(step, 3100, step), rnd.randrange(step, 3100, step), rnd.randrange(8, 130)
        if step == 8 and rnd.random() < 0.3:  # whole large tiles: what a split on large tiles asks for
            m, n = max(256, m // 256 * 256), max(256, n // 256 * 256)
        # bound the work of one case (operands are made in fp32 first)
        while m * n * kt > 4096 * 4096 * 240:
            kt = max(1, kt // 2)
        return kind, m, n, kt

    with open(a.out, "w") as out:
        for case in range(a.cases):
            dtype = rnd.choice(["bf16", "fp8", "mxfp8"])
            kind, m, n, kt = shape(dtype)
            k = kt * (64 if dtype == "bf16" else 128)
            if a.soak > 0:
                schedule, splits = diag.gemm_plan(dtype, m, n, k)
                row = {"case": case, "kind": kind, "dtype": dtype, "m": m, "n": n, "k": k, "schedule": schedule, "splits": splits}
                try:
                    r = diag.gemm_stress(dtype, m, n, k, seconds=a.soak, verify=True, check_every=1)
                    row.update(launches=r["launches"], checked_in_kernel=r["checked_in_kernel"], checked_in_memory=r["checked_in_memory"],
                               rel_err=r["max_rel_err"], sdc_mismatches=r["sdc_mismatches"])
                    row["ok"] = (r["verified"] == 1 and r["sdc_mismatches"] == 0 and r["launches"] > 0 and r["max_rel_err"] < 6e-3
                                 and r["checked_in_kernel"] == r["launches"] and r["checked_in_memory"] > 0)
                except diag.DiagError as e:
                    row.update(ok=False, error=str(e)[-300:])
                out.write(json.dumps(row) + "\n")
                out.flush()
                key = f"{dtype} schedule {schedule}" + (f" x{splits}" if splits > 1 else "")
                by_schedule[key] = by_schedule.get(key, 0) + 1
                if not row["ok"]:
                    failed += 1
                    print("FAILED:", json.dumps(row), flush=True)
                if time.time() - t_last > 20:
                    print(f"{case + 1} of {a.cases} cases, {failed} failed", flush=True)
                    t_last = time.time()
                continue
            g = torch.Generator(device=dev).manual_seed(a.seed * 100003 + case)
            af, bf = torch.randn(m, k, device=dev, generator=g), torch.randn(n, k, device=dev, generator=g)
            kw = {}
            if dtype == "bf16":
                qa, qb = af.bfloat16(), bf.bfloat16()
                da, db = qa.float(), qb.float()
            else:
                qa, qb = af.to(torch.float8_e4m3fn), bf.to(torch.float8_e4m3fn)
                da, db = qa.float(), qb.float()
            if dtype == "mxfp8":
                sa = torch.randint(122, 133, (m, k // 32), dtype=torch.uint8, device=dev, generator=g)
                sb = torch.randint(122, 133, (n, k // 32), dtype=torch.uint8, device=dev, generator=g)
                pa, pb = torch.empty_like(sa), torch.empty_like(sb)
                diag.mx_pack_scales(sa.data_ptr(), pa.data_ptr(), m, k, stream)
                diag.mx_pack_scales(sb.data_ptr(), pb.data_ptr(), n, k, stream)
                da = da * torch.exp2(sa.float() - 127.0).repeat_interleave(32, dim=1)
                db = db * torch.exp2(sb.float() - 127.0).repeat_interleave(32, dim=1)
                kw = {"scale_a": pa.data_ptr(), "scale_b": pb.data_ptr()}
            del af, bf
            ref = da @ db.t()
            del da, db
            schedule, splits = diag.gemm_plan(dtype, m, n, k)
            guard = 4096
            buf = torch.full((m * n + guard,), 7.0, dtype=torch.bfloat16, device=dev)
            c = buf[:m * n].view(m, n)
            diag.gemm_launch(dtype, qa.data_ptr(), qb.data_ptr(), c.data_ptr(), m, n, k, stream, **kw)
            c2 = torch.empty_like(c)
            diag.gemm_launch(dtype, qa.data_ptr(), qb.data_ptr(), c2.data_ptr(), m, n, k, stream, **kw)
            torch.cuda.synchronize()
            finite = bool(torch.isfinite(c.float()).all())
            err = ((c.float() - ref).abs().max() / ref.abs().max().clamp_min(1e-6)).item() if finite else float("inf")
            row = {"case": case, "kind": kind, "dtype": dtype, "m": m, "n": n, "k": k, "schedule": schedule, "splits": splits,
                   "rel_err": err, "guard_ok": bool((buf[m * n:] == 7.0).all()), "repeatable": bool(torch.equal(c, c2))}
            row["ok"] = finite and err < 6e-3 and row["guard_ok"] and row["repeatable"]
            out.write(json.dumps(row) + "\n")
            out.flush()
            key = f"{dtype} schedule {schedule}" + (f" x{splits}" if splits > 1 else "")
            by_schedule[key] = by_schedule.get(key, 0) + 1
            if not row["ok"]:
                failed += 1
                print("FAILED:", json.dumps(row), flush=True)
            if time.time() - t_last > 20:
                print(f"{case + 1} of {a.cases} cases, {failed} failed", flush=True)
                t_last = time.time()
            del ref, buf, c, c2, qa, qb
    print(json.dumps({"cases": a.cases, "seed": a.seed, "soak_s": a.soak, "failed": failed, "by_schedule": dict(sorted(by_schedule.items()))}), flush=True)
    return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
