"""Interconnect validation: RCCL collective and point-to-point bandwidth sweeps.

One process per GPU over ``torch.distributed`` (backend "nccl" is RCCL on ROCm,
"gloo" on CPU for tests).  MI355X nodes wire the 8 GPUs as a full mesh of xGMI
links (7 per GPU) rather than through a switch, so a ring collective is bound by
ONE link per hop while all-to-all and the pairwise matrix can drive all 7 at
once; the sweep therefore reports, for every collective and size, both algorithm
bandwidth and bus bandwidth, and the pairwise matrix exposes a single weak link
that a ring number would only show as a uniformly slower ring.

Bus-bandwidth factors (bytes that cross the busiest link per payload byte):
  all_reduce      2 (n-1) / n
  all_gather        (n-1) / n      payload = the gathered (total) buffer
  reduce_scatter    (n-1) / n      payload = the scattered (total) buffer
  all_to_all        (n-1) / n      payload = one rank's send buffer
  broadcast       1

The reference has no interconnect test; its closest feature is the topology
report (bindings/go/dcgm/topology.go:74-129), which names links but never
exercises them.
"""
import math
import os
import time
from dataclasses import asdict, dataclass

import torch
import torch.distributed as dist

OPS = ("all_reduce", "all_gather", "reduce_scatter", "all_to_all", "broadcast")


def busbw_factor(op, n):
    if n <= 1:
        return 1.0
    if op == "all_reduce":
        return 2.0 * (n - 1) / n
    if op in ("all_gather", "reduce_scatter", "all_to_all"):
        return (n - 1) / n
    if op == "broadcast":
        return 1.0
    raise ValueError(op)


@dataclass
class SweepPoint:
    op: str
    bytes: int          # payload bytes as defined in the module docstring
    world: int
    iters: int
    time_us: float      # mean per-iteration time, max over ranks
    algbw_gbps: float   # payload / time, GB/s (1e9)
    busbw_gbps: float
    verified: bool

    def as_dict(self):
        return asdict(self)


def default_sizes(lo=8, hi=1 << 30, factor=4):
    """8 B ... 1 GiB in x4 steps; the last size is always `hi`."""
    s, out = lo, []
    while s < hi:
        out.append(s)
        s *= factor
    out.append(hi)
    return out


def _sync(device):
    if device.type == "cuda":
        torch.cuda.synchronize(device)


def _elem(dtype):
    return torch.empty((), dtype=dtype).element_size()


class _Case:
    """Buffers + launch + integrity check for one (op, payload size)."""

    def __init__(self, op, nbytes, dtype, device, rank, world):
        self.op, self.rank, self.world, self.device, self.dtype = op, rank, world, device, dtype
        es = _elem(dtype)
        per_rank_ops = op in ("all_gather", "reduce_scatter", "all_to_all")
        # element count, rounded so that it splits evenly across ranks where the op needs it
        n = max(1, nbytes // es)
        if per_rank_ops:
            n = max(world, (n // world) * world)
        self.n = n
        self.bytes = n * es
        shard = n // world if per_rank_ops else n
        # Small integers so that sums are exact in bf16/fp16 for any world <= 16.
        base = (torch.arange(n, device=device) % 7).to(dtype)
        if op == "all_reduce":
            self.buf = base + rank
            self.src = self.buf.clone()
        elif op == "broadcast":
            self.buf = base + rank
        elif op == "all_gather":
            self.out = torch.empty(n, dtype=dtype, device=device)
            self.inp = base[rank * shard:(rank + 1) * shard] + rank
        elif op == "reduce_scatter":
            self.inp = base + rank
            self.out = torch.empty(shard, dtype=dtype, device=device)
        elif op == "all_to_all":
            self.inp = base + rank
            self.out = torch.empty(n, dtype=dtype, device=device)
        self.shard = shard
        self.base = base

    def launch(self):
        op = self.op
        if op == "all_reduce":
            dist.all_reduce(self.buf)
        elif op == "broadcast":
            dist.broadcast(self.buf, src=0)
        elif op == "all_gather":
            dist.all_gather_into_tensor(self.out, self.inp)
        elif op == "reduce_scatter":
            dist.reduce_scatter_tensor(self.out, self.inp)
        elif op == "all_to_all":
            dist.all_to_all_single(self.out, self.inp)

    def verify(self):
        """One checked run on freshly initialised data: a corrupting link shows as a mismatch."""
        w, r, sh = self.world, self.rank, self.shard
        f32 = lambda t: t.to(torch.float32)
        if self.op == "all_reduce":
            self.buf.copy_(self.src)
            self.launch()
            want = f32(self.base) * w + w * (w - 1) / 2
            ok = torch.equal(f32(self.buf), want)
            self.buf.copy_(self.src)
        elif self.op == "broadcast":
            self.launch()
            ok = torch.equal(f32(self.buf), f32(self.base))
        elif self.op == "all_gather":
            self.out.zero_()
            self.launch()
            owner = torch.arange(self.n, device=self.device) // sh
            ok = torch.equal(f32(self.out), f32(self.base) + owner)
        elif self.op == "reduce_scatter":
            self.launch()
            want = f32(self.base[r * sh:(r + 1) * sh]) * w + w * (w - 1) / 2
            ok = torch.equal(f32(self.out), want)
        else:  # all_to_all: chunk j of my output is chunk `rank` of rank j's input
            self.out.zero_()
            self.launch()
            owner = torch.arange(self.n, device=self.device) // sh
            mine = f32(self.base[r * sh:(r + 1) * sh]).repeat(w)
            ok = torch.equal(f32(self.out), mine + owner)
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())


def _time_case(case, iters, warmup, device):
    for _ in range(warmup):
        case.launch()
    _sync(device)
    dist.barrier()
    _sync(device)
    t0 = time.perf_counter()
    for _ in range(iters):
        case.launch()
    _sync(device)
    dt = (time.perf_counter() - t0) / iters
    t = torch.tensor([dt], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def _iters_for(nbytes, iters):
    """Fewer repetitions for the big sizes: keeps a full sweep within seconds."""
    if iters:
        return iters
    if nbytes <= 1 << 20:
        return 50
    if nbytes <= 64 << 20:
        return 20
    return 8


def sweep(ops=OPS, sizes=None, dtype=torch.bfloat16, device=None, iters=0, warmup=3, verify=True):
    """Runs on every rank; returns the same list of SweepPoint on all of them.

    Ops the backend cannot run (gloo lacks some) are skipped, not failed."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device(device) if device is not None else _default_device()
    sizes = sizes or default_sizes()
    out = []
    for op in ops:
        for nbytes in sizes:
            try:
                case = _Case(op, nbytes, dtype, device, rank, world)
                ok = case.verify() if verify else True
                n_it = _iters_for(case.bytes, iters)
                dt = _time_case(case, n_it, warmup, device)
            except (RuntimeError, NotImplementedError) as e:
                if "not supported" in str(e).lower() or "does not support" in str(e).lower() or isinstance(e, NotImplementedError):
                    break
                raise
            alg = case.bytes / dt / 1e9
            out.append(SweepPoint(op, case.bytes, world, n_it, dt * 1e6, alg, alg * busbw_factor(op, world), ok))
            del case
    return out


def _default_device():
    if dist.get_backend() == "nccl":
        return torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0)))
    return torch.device("cpu")


def p2p_matrix(nbytes=256 << 20, dtype=torch.bfloat16, device=None, iters=5, bidirectional=False):
    """Pairwise bandwidth matrix in GB/s: entry [i][j] is i -> j with every other rank idle.

    On MI355X every ordered pair maps to one xGMI link, so a degraded link is the
    one low cell; collectives cannot localise it.  Returned on every rank."""
    rank, world = dist.get_rank(), dist.get_world_size()
    device = torch.device(device) if device is not None else _default_device()
    n = max(1, nbytes // _elem(dtype))
    send = torch.ones(n, dtype=dtype, device=device)
    recv = torch.empty(n, dtype=dtype, device=device)
    m = torch.zeros(world, world, dtype=torch.float64, device=device)
    for i in range(world):
        for j in range(world):
            if i == j or (bidirectional and j < i):
                continue
            dist.barrier()
            if rank in (i, j):
                def once():
                    ops = []
                    if rank == i:
                        ops.append(dist.P2POp(dist.isend, send, j))
                        if bidirectional:
                            ops.append(dist.P2POp(dist.irecv, recv, j))
                    else:
                        ops.append(dist.P2POp(dist.irecv, recv, i))
                        if bidirectional:
                            ops.append(dist.P2POp(dist.isend, send, i))
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                once()  # connection setup + warmup
                _sync(device)
                t0 = time.perf_counter()
                for _ in range(iters):
                    once()
                _sync(device)
                dt = (time.perf_counter() - t0) / iters
                if rank == j:  # the receiver's clock covers the whole transfer
                    gbps = n * _elem(dtype) * (2 if bidirectional else 1) / dt / 1e9
                    m[i, j] = gbps
                    if bidirectional:
                        m[j, i] = gbps
    dist.all_reduce(m, op=dist.ReduceOp.MAX)
    return m.cpu().tolist()


def format_table(points):
    """nccl-tests style table."""
    lines = [f"# {'op':<15}{'bytes':>14}{'iters':>7}{'time(us)':>12}{'algbw(GB/s)':>13}{'busbw(GB/s)':>13}  check"]
    for p in points:
        lines.append(f"  {p.op:<15}{p.bytes:>14}{p.iters:>7}{p.time_us:>12.1f}{p.algbw_gbps:>13.2f}{p.busbw_gbps:>13.2f}  "
                     f"{'ok' if p.verified else 'MISMATCH'}")
    return "\n".join(lines)


def peak_busbw(points):
    """{op: best busbw over sizes} — the number quoted per collective."""
    best = {}
    for p in points:
        if not math.isnan(p.busbw_gbps):
            best[p.op] = max(best.get(p.op, 0.0), p.busbw_gbps)
    return best
