"""Interconnect sweep on the gloo backend with real multi-process rendezvous
(world_size 2 and 3 on 127.0.0.1): bus-bandwidth math, data-integrity checks
that catch a corrupted collective, the pairwise matrix and the CLI."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from k8s_gpu_monitor_amd.parallel import collectives as C
from k8s_gpu_monitor_amd.parallel.sweep import parse_size

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_busbw_factors_and_sizes():
    assert C.busbw_factor("all_reduce", 8) == pytest.approx(1.75)
    assert C.busbw_factor("all_gather", 8) == pytest.approx(0.875)
    assert C.busbw_factor("all_to_all", 2) == pytest.approx(0.5)
    assert C.busbw_factor("reduce_scatter", 4) == pytest.approx(0.75)
    assert C.busbw_factor("broadcast", 8) == 1.0
    assert all(C.busbw_factor(op, 1) == 1.0 for op in C.OPS)
    with pytest.raises(ValueError):
        C.busbw_factor("gossip", 4)
    s = C.default_sizes()
    assert s[0] == 8 and s[-1] == 1 << 30 and all(b == a * 4 for a, b in zip(s[:-2], s[1:-1]))
    assert parse_size("1G") == 1 << 30 and parse_size("256m") == 256 << 20 and parse_size("4096") == 4096


def _worker(rank, world, port, q, corrupt):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        if corrupt and rank == 1:
            # A link that flips data: rank 1 contributes something other than what the check expects.
            real = C._Case.launch

            def bad(self):
                if self.op == "all_reduce":
                    self.buf[0] += 1
                real(self)
            C._Case.launch = bad
        pts = C.sweep(sizes=[64, 4096, 1 << 20], dtype=torch.float32, iters=3, warmup=1)
        m = C.p2p_matrix(1 << 20, torch.float32, iters=2)
        if rank == 0:
            q.put(([p.as_dict() for p in pts], m))
    finally:
        dist.destroy_process_group()


def _run(world, corrupt=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, corrupt)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    return out


@pytest.mark.parametrize("world", [2, 3])
def test_sweep_gloo(world):
    pts, m = _run(world)
    ops = {p["op"] for p in pts}
    assert {"all_reduce", "all_gather", "all_to_all", "broadcast"} <= ops
    for p in pts:
        assert p["verified"], p
        assert p["world"] == world and p["time_us"] > 0
        assert p["busbw_gbps"] == pytest.approx(p["algbw_gbps"] * C.busbw_factor(p["op"], world))
        assert p["bytes"] % 4 == 0
        if p["op"] in ("all_gather", "all_to_all", "reduce_scatter"):
            assert (p["bytes"] // 4) % world == 0  # payload splits evenly
    assert len(m) == world
    for i in range(world):
        for j in range(world):
            assert (m[i][j] == 0) if i == j else (m[i][j] > 0)


def test_sweep_detects_corruption():
    pts, _ = _run(2, corrupt=True)
    bad = [p for p in pts if not p["verified"]]
    assert bad and all(p["op"] == "all_reduce" for p in bad)
    assert all(p["verified"] for p in pts if p["op"] != "all_reduce")


def test_cli_under_torchrun(tmp_path):
    out = tmp_path / "sweep.json"
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_port()), "-m",
                        "k8s_gpu_monitor_amd.parallel.sweep", "--backend", "gloo", "--max-bytes", "64K", "--dtype", "fp32",
                        "--p2p", "--p2p-bytes", "64K", "--json", str(out)], capture_output=True, text=True, timeout=300,
                       env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "# peak busbw (GB/s):" in r.stdout and "pairwise GB/s" in r.stdout
    d = json.loads(out.read_text())
    assert d["world"] == 2 and d["backend"] == "gloo" and len(d["p2p_gbps"]) == 2
    assert {p["op"] for p in d["points"]} >= {"all_reduce", "all_gather"}
