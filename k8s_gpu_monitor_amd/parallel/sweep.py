"""Command-line entry of the interconnect sweep; one rank per GPU.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        -m k8s_gpu_monitor_amd.parallel.sweep --max-bytes 1G --json out.json

Single-process use (``python -m k8s_gpu_monitor_amd.parallel.sweep``) runs the
1-GPU point of the 1/2/4/8 scaling series: collectives degenerate to local
copies, which is the on-device ceiling the multi-GPU points are compared with.
"""
import argparse
import datetime
import json
import os
import sys

import torch
import torch.distributed as dist

from . import collectives as C


def parse_size(s):
    s = s.strip().upper()
    mult = {"K": 1 << 10, "M": 1 << 20, "G": 1 << 30}
    if s and s[-1] in mult:
        return int(float(s[:-1]) * mult[s[-1]])
    return int(s)


def init_distributed(backend=None, timeout_s=300):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun's contract); defaults to a 1-rank job."""
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    # The host driver only offers dmabuf IPC; RCCL's peer mapping fails without this.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    kw = {}
    if backend == "nccl":
        torch.cuda.set_device(local)
        kw["device_id"] = torch.device("cuda", local)
    dist.init_process_group(backend, rank=rank, world_size=world, timeout=datetime.timedelta(seconds=timeout_s), **kw)
    return rank, world, local


def main(argv=None):
    ap = argparse.ArgumentParser(description="RCCL collective + pairwise bandwidth sweep over xGMI")
    ap.add_argument("--ops", default=",".join(C.OPS))
    ap.add_argument("--min-bytes", default="8")
    ap.add_argument("--max-bytes", default="1G")
    ap.add_argument("--factor", type=int, default=4)
    ap.add_argument("--iters", type=int, default=0, help="0 = size-dependent")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp16", "fp32"])
    ap.add_argument("--backend", default=None)
    ap.add_argument("--p2p", action="store_true", help="also measure the pairwise link matrix")
    ap.add_argument("--p2p-bytes", default="256M")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--json", default=None, help="write results here (rank 0)")
    a = ap.parse_args(argv)
    rank, world, _ = init_distributed(a.backend)
    dtype = {"bf16": torch.bfloat16, "fp16": torch.float16, "fp32": torch.float32}[a.dtype]
    try:
        sizes = C.default_sizes(parse_size(a.min_bytes), parse_size(a.max_bytes), a.factor)
        pts = C.sweep([o for o in a.ops.split(",") if o], sizes, dtype, iters=a.iters, verify=not a.no_verify)
        matrix = C.p2p_matrix(parse_size(a.p2p_bytes), dtype) if a.p2p and world > 1 else None
        bad = [p for p in pts if not p.verified]
        if rank == 0:
            print(f"# world {world}  backend {dist.get_backend()}  dtype {a.dtype}")
            print(C.format_table(pts))
            print("# peak busbw (GB/s): " + "  ".join(f"{k}={v:.1f}" for k, v in C.peak_busbw(pts).items()))
            if matrix:
                print("# pairwise GB/s (row = sender)")
                for row in matrix:
                    print("  " + " ".join(f"{x:8.1f}" for x in row))
            if a.json:
                with open(a.json, "w") as f:
                    json.dump({"world": world, "backend": dist.get_backend(), "dtype": a.dtype,
                               "points": [p.as_dict() for p in pts], "peak_busbw_gbps": C.peak_busbw(pts),
                               "p2p_gbps": matrix}, f, indent=1)
        return 1 if bad else 0
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    sys.exit(main())
