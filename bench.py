#!/usr/bin/env python3
"""Headline benchmark: one active-diagnostics step per GPU while the node exporter
monitors every GPU at 1 s and is scraped over HTTP.

A step, on every rank (= one MI355X), runs the whole diag suite once:
  hbm      read / write / copy / triad STREAM kernels over 4 GiB buffers
  memtest  one pattern fill + recompute-verify over a 4 GiB buffer
  bf16     GEMM 8192^3 on the MFMA kernel, G launches, random data
  mxfp8    GEMM 8192^3 on the scaled-MFMA kernel, G launches, random data
  rccl     all-reduce / all-gather / all-to-all of 1 GiB over xGMI   (N > 1)
  scrape   rank 0 fetches /metrics from the gpumon-exporter child process
The exporter child is the production binary with the amdsmi backend, watching all
of the node's GPUs at 1 s with profiling + AMD extra families, the health check and the
policy registration (gpumon_health_*, gpumon_policy_violations_total); its CPU time is
read from /proc over a window of at least ten seconds.

`value` is the aggregate bf16 GEMM rate of the job (sum over ranks); the other
headline numbers (HBM TB/s, MX-fp8 TFLOPS, RCCL bus bandwidth, scrape p50,
exporter CPU %) are in `extras`.  Nothing is skipped in the timed region and
per-phase times come from HIP events on the stream the phases run on.

`--dry-run` exercises the same control flow (rank handling, collectives, the
reductions, the JSON) on CPU tensors over gloo with toy sizes; it exists so the
multi-rank path can be tested without GPUs and its output is labelled as such.
"""
import argparse
import http.client
import json
import os
import re
import signal
import statistics
import subprocess
import sys
import threading
import time
import urllib.request

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

MEMTEST_PATTERN = "random"


class Sizes:
    def __init__(self, dry):
        self.gemm_dim = 256 if dry else 8192
        self.hbm_bytes = (1 << 20) if dry else (4 << 30)
        self.coll_bytes = (1 << 20) if dry else (1 << 30)


SCRAPE_ERRORS = (OSError, ValueError, http.client.HTTPException)  # what a dead or wedged exporter makes urllib raise
EXPORTER_WINDOW_S = 10.0  # shortest window the exporter's CPU share is measured over


class ExporterChild:
    """gpumon-exporter as a child process (started before this process touches the GPU)."""

    def __init__(self, backend="amdsmi", prof_counters=False, interval_ms=1000, extra_args=()):
        from k8s_gpu_monitor_amd._lib import binary
        env = dict(os.environ, GPUMON_BACKEND=backend)
        if prof_counters:  # MFMA pipe activity from hardware counters (libgpumon_prof.so inside the exporter)
            env["GPUMON_PROF_COUNTERS"] = "1"
        if backend == "fake":
            env.setdefault("GPUMON_BACKEND_SPEC", "gpus=8,seed=1")
        self.proc = subprocess.Popen(
            [binary("gpumon-exporter"), "-e", "-p", "-o", "", "-d", str(int(interval_ms)), "--extras", "--self-metrics", "--health",
             "--listen", "127.0.0.1:0", *extra_args], stderr=subprocess.PIPE, text=True, env=env)
        self.port = None
        self.log = []
        deadline = time.time() + 60
        while time.time() < deadline:
            line = self.proc.stderr.readline()
            if not line:
                break
            self.log.append(line.rstrip())
            m = re.search(r"Serving /metrics on port (\d+)", line)
            if m:
                self.port = int(m.group(1))
                break
        if self.port is None:
            self.stop()
            raise RuntimeError("gpumon-exporter did not come up:\n" + "\n".join(self.log))
        threading.Thread(target=lambda: [self.log.append(l.rstrip()) for l in self.proc.stderr], daemon=True).start()
        self.url = f"http://127.0.0.1:{self.port}/metrics"
        self.tck = os.sysconf("SC_CLK_TCK")

    def cpu_seconds(self):
        """On-CPU time of all the child's threads: schedstat is in ns, /proc/stat ticks are 10 ms."""
        total, ok = 0, False
        try:
            for t in os.listdir(f"/proc/{self.proc.pid}/task"):
                with open(f"/proc/{self.proc.pid}/task/{t}/schedstat") as f:
                    total += int(f.read().split()[0])
                    ok = True
        except OSError:
            ok = False
        if ok:
            return total / 1e9
        try:
            with open(f"/proc/{self.proc.pid}/stat") as f:
                rest = f.read().rsplit(")", 1)[1].split()
        except OSError:
            return None  # the child is gone
        return (int(rest[11]) + int(rest[12])) / self.tck  # utime + stime

    def rss_kb(self):
        try:
            with open(f"/proc/{self.proc.pid}/status") as f:
                for line in f:
                    if line.startswith("VmRSS:"):
                        return int(line.split()[1])
        except OSError:
            pass
        return 0

    def scrape(self):
        t0 = time.perf_counter()
        body = urllib.request.urlopen(self.url, timeout=10).read()
        return (time.perf_counter() - t0) * 1e3, body

    def stop(self):
        if self.proc.poll() is None:
            self.proc.send_signal(signal.SIGTERM)
            try:
                self.proc.wait(timeout=15)
            except subprocess.TimeoutExpired:
                self.proc.kill()


class GpuKernels:
    """The seven compute phases on one MI355X: libgpumon_diag kernels on torch-owned buffers."""

    def __init__(self, torch, dev, sz, G, rank):
        from k8s_gpu_monitor_amd.ops import diag
        self.torch, self.diag, self.dev, self.sz, self.G = torch, diag, dev, sz, G
        self.stream = torch.cuda.current_stream(dev)
        self.sptr = self.stream.cuda_stream
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        D = sz.gemm_dim
        self.nf = sz.hbm_bytes // 4
        self.ha = torch.empty(self.nf, dtype=torch.float32, device=dev).normal_(generator=g)
        self.hb = torch.empty(self.nf, dtype=torch.float32, device=dev).normal_(generator=g)
        self.hc = torch.empty(self.nf, dtype=torch.float32, device=dev)
        self.sink = torch.zeros(1 << 16, dtype=torch.float32, device=dev)
        self.A16 = torch.empty(D, D, dtype=torch.bfloat16, device=dev).normal_(generator=g)
        self.B16 = torch.empty(D, D, dtype=torch.bfloat16, device=dev).normal_(generator=g)
        self.C16 = torch.empty(D, D, dtype=torch.bfloat16, device=dev)
        self.A8 = torch.empty(D, D, dtype=torch.float32, device=dev).normal_(generator=g).to(torch.float8_e4m3fn)
        self.B8 = torch.empty(D, D, dtype=torch.float32, device=dev).normal_(generator=g).to(torch.float8_e4m3fn)
        self.C8 = torch.empty(D, D, dtype=torch.bfloat16, device=dev)
        # MX block scales: one E8M0 byte per row and 32 k, 2^-3 .. 2^3, in the packed layout the kernel reads
        self.SA8, self.SB8 = (torch.empty(D, D // 32, dtype=torch.uint8, device=dev) for _ in range(2))
        for packed in (self.SA8, self.SB8):
            nat = torch.randint(124, 131, (D, D // 32), dtype=torch.uint8, device=dev, generator=g)
            diag.mx_pack_scales(nat.data_ptr(), packed.data_ptr(), D, D, self.sptr)
        torch.cuda.synchronize()

    def event(self):
        return self.torch.cuda.Event(enable_timing=True)

    def record(self, e):
        e.record(self.stream)

    def elapsed_ms(self, e0, e1):
        return e0.elapsed_time(e1)

    def sync(self):
        self.torch.cuda.synchronize(self.dev)

    def comm_tensor(self, n):
        g = self.torch.Generator(device=self.dev).manual_seed(99)
        return self.torch.empty(n, dtype=self.torch.bfloat16, device=self.dev).normal_(generator=g)

    def hbm_read(self):
        self.diag.hbm_stream_launch("read", self.ha.data_ptr(), 0, self.sink.data_ptr(), 0.0, self.nf, "auto", self.sptr)

    def hbm_write(self):
        self.diag.hbm_stream_launch("write", self.hc.data_ptr(), 0, 0, 1.5, self.nf, "auto", self.sptr)

    def hbm_copy(self):
        self.diag.hbm_stream_launch("copy", self.ha.data_ptr(), 0, self.hc.data_ptr(), 0.0, self.nf, "auto", self.sptr)

    def hbm_triad(self):
        self.diag.hbm_stream_launch("triad", self.ha.data_ptr(), self.hb.data_ptr(), self.hc.data_ptr(), 0.5, self.nf, "auto",
                                    self.sptr)

    def memtest(self, i):
        self.diag.memtest_fill(self.hc.data_ptr(), self.sz.hbm_bytes // 8, MEMTEST_PATTERN, seed=i, stream=self.sptr)
        errs, _ = self.diag.memtest_verify(self.hc.data_ptr(), self.sz.hbm_bytes // 8, MEMTEST_PATTERN, seed=i, stream=self.sptr)
        return errs

    def gemm_bf16(self):
        D = self.sz.gemm_dim
        for _ in range(self.G):
            self.diag.gemm_launch("bf16", self.A16.data_ptr(), self.B16.data_ptr(), self.C16.data_ptr(), D, D, D, self.sptr)

    def gemm_mxfp8(self):
        D = self.sz.gemm_dim
        for _ in range(self.G):
            self.diag.gemm_launch("mxfp8", self.A8.data_ptr(), self.B8.data_ptr(), self.C8.data_ptr(), D, D, D, self.sptr,
                                  scale_a=self.SA8.data_ptr(), scale_b=self.SB8.data_ptr())


class DryKernels:
    """CPU stand-ins with the same interface (toy sizes): for testing the control flow only."""

    class _Ev:
        t = 0.0

    def __init__(self, torch, dev, sz, G, rank):
        self.torch, self.dev, self.sz, self.G = torch, dev, sz, G
        D = sz.gemm_dim
        self.x = torch.randn(sz.hbm_bytes // 4)
        self.y = torch.randn(sz.hbm_bytes // 4)
        self.a = torch.randn(D, D)

    def event(self):
        return DryKernels._Ev()

    def record(self, e):
        e.t = time.perf_counter()

    def elapsed_ms(self, e0, e1):
        return max((e1.t - e0.t) * 1e3, 1e-6)

    def sync(self):
        pass

    def comm_tensor(self, n):
        return self.torch.randn(n)  # fp32: gloo's reductions do not cover bf16 on every build

    def hbm_read(self):
        self.x.sum()

    def hbm_write(self):
        self.y.fill_(1.5)

    def hbm_copy(self):
        self.y.copy_(self.x)

    def hbm_triad(self):
        self.torch.add(self.x, self.y, alpha=0.5, out=self.y)

    def memtest(self, i):
        return 0

    def gemm_bf16(self):
        for _ in range(self.G):
            self.a @ self.a

    gemm_mxfp8 = gemm_bf16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--gemm-launches", type=int, default=8, help="GEMM launches per dtype per step")
    ap.add_argument("--no-exporter", action="store_true", help="debug: skip the exporter child and the scrape phase")
    ap.add_argument("--prof-counters", action="store_true",
                    help="the exporter also programs the MFMA busy counter (rocprofiler-sdk device counting) on every GPU")
    ap.add_argument("--kill-exporter-at", type=int, default=-1, help=argparse.SUPPRESS)  # test hook: SIGKILL the child before step N
    ap.add_argument("--dry-run", action="store_true", help="CPU + gloo + fake backend with toy sizes: control-flow test only")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    local = int(os.environ.get("LOCAL_RANK", 0))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit(f"--gpus {a.gpus} needs a launcher: python -m torch.distributed.run --nnodes=1 --nproc-per-node "
                     f"{a.gpus} --master-addr 127.0.0.1 bench.py --gpus {a.gpus} ...")
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    exporter, exporter_error = None, None
    if rank == 0 and not a.no_exporter:
        try:
            exporter = ExporterChild("fake" if a.dry_run else "amdsmi", a.prof_counters)  # before any HIP call in this process
        except (OSError, RuntimeError) as e:
            # The kernels are still measured; the JSON says that the monitoring half is missing and why.
            exporter_error = str(e)[-400:]
            print(f"bench: exporter not started: {exporter_error}", file=sys.stderr)

    import torch
    import torch.distributed as dist
    from k8s_gpu_monitor_amd.parallel.collectives import busbw_factor

    sz = Sizes(a.dry_run)
    if a.dry_run:
        dev = torch.device("cpu")
        backend, kw = "gloo", {}
    else:
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
        backend, kw = "nccl", {"device_id": dev}
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)

    try:
        # The communicator must span exactly N ranks on N different devices: a launcher that put two ranks on one GPU,
        # or a communicator that came up smaller, would still produce numbers.
        ranks_seen, devices_seen = 1, 1
        if world > 1:
            if a.dry_run:
                ident = f"cpu-rank{rank}"
            else:
                props = torch.cuda.get_device_properties(dev)
                ident = "/".join(str(getattr(props, k, "?")) for k in ("uuid", "pci_domain_id", "pci_bus_id", "pci_device_id"))
            seen = [None] * world
            dist.all_gather_object(seen, (rank, ident))
            ranks_seen, devices_seen = len({r for r, _ in seen}), len({d for _, d in seen})
            if dist.get_world_size() != world or ranks_seen != world:
                sys.exit(f"bench: asked for {world} ranks, the communicator has {dist.get_world_size()} ({ranks_seen} distinct)")
            if devices_seen != world and rank == 0:
                # Recorded rather than fatal: the identity comes from the runtime's device properties, and a runtime that
                # reports the same identity for every card must not stop a correct job.
                print(f"bench: {world} ranks report {devices_seen} distinct device identities: {sorted(d for _, d in seen)}",
                      file=sys.stderr)
        kernels = (DryKernels if a.dry_run else GpuKernels)(torch, dev, sz, a.gemm_launches, rank)
        result = run(a, world, dev, exporter, torch, dist, kernels, sz, busbw_factor)
        result["extras"]["collective_ranks"] = ranks_seen
        result["extras"]["collective_devices"] = devices_seen
        if exporter_error:
            result["extras"]["exporter_error"] = exporter_error
    finally:
        if exporter:
            exporter.stop()
        if world > 1:
            dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


def run(a, world, dev, exporter, torch, dist, K, sz, busbw_factor):
    G, D = a.gemm_launches, sz.gemm_dim
    if world > 1:
        ne = sz.coll_bytes // 2
        ar = K.comm_tensor(ne)
        ag_in = ar[: ne // world]
        ag_out = torch.empty(ne, dtype=ar.dtype, device=dev)
        a2a_out = torch.empty(ne, dtype=ar.dtype, device=dev)

    phases = ["hbm_read", "hbm_write", "hbm_copy", "hbm_triad", "memtest", "gemm_bf16", "gemm_mxfp8"]
    if world > 1:
        phases += ["all_reduce", "all_gather", "all_to_all"]
    total_steps = a.warmup + a.steps
    ev = [[K.event() for _ in range(len(phases) + 1)] for _ in range(total_steps)]
    scrape_ms, scrape_bytes, scrape_failures = [], 0, []
    memtest_errors = 0

    def step(i):
        nonlocal memtest_errors, scrape_bytes
        e = ev[i]
        K.record(e[0])
        K.hbm_read()
        K.record(e[1])
        K.hbm_write()
        K.record(e[2])
        K.hbm_copy()
        K.record(e[3])
        K.hbm_triad()
        K.record(e[4])
        memtest_errors += K.memtest(i)
        K.record(e[5])
        K.gemm_bf16()
        K.record(e[6])
        K.gemm_mxfp8()
        K.record(e[7])
        if world > 1:
            dist.all_reduce(ar)
            K.record(e[8])
            dist.all_gather_into_tensor(ag_out, ag_in)
            K.record(e[9])
            dist.all_to_all_single(a2a_out, ar)
            K.record(e[10])
            if i % 8 == 7:  # keep the all-reduced values bounded
                ar.mul_(1.0 / world ** 8)
        if exporter is not None and i == a.kill_exporter_at:
            exporter.proc.kill()
            exporter.proc.wait()
        if exporter is not None and not scrape_failures:
            # A monitoring process that dies or stops answering must not take the kernels' measurement with it: the
            # first failure ends the scrape phase for the rest of the run and is reported in the JSON.
            try:
                ms, body = exporter.scrape()
            except SCRAPE_ERRORS as e:
                scrape_failures.append(f"step {i}: {e}"[-300:])
                print(f"bench: scrape failed at step {i}: {e}", file=sys.stderr)
            else:
                if i >= a.warmup:
                    scrape_ms.append(ms)
                    scrape_bytes = len(body)

    def fence():
        K.sync()
        if world > 1:
            dist.barrier()
        K.sync()

    # Exporter cost is a ratio of two small numbers at a 1 s sampling period: it is measured over a window of its own
    # that starts here and lasts at least EXPORTER_WINDOW_S, however short the timed region is.
    cpu_w0, t_w0 = (exporter.cpu_seconds(), time.perf_counter()) if exporter else (0.0, 0.0)
    for i in range(a.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(a.warmup, total_steps):
        step(i)
    fence()
    elapsed = time.perf_counter() - t0
    exporter_window_s, exporter_cpu_s = 0.0, 0.0
    if exporter:  # outside the timed region: keep scraping once a second until the window is long enough
        while time.perf_counter() - t_w0 < EXPORTER_WINDOW_S:
            time.sleep(min(1.0, max(0.0, EXPORTER_WINDOW_S - (time.perf_counter() - t_w0))))
            try:
                exporter.scrape()
            except SCRAPE_ERRORS:
                break
        exporter_window_s = time.perf_counter() - t_w0
        cpu_w1 = exporter.cpu_seconds()
        exporter_cpu_s = cpu_w1 - cpu_w0 if cpu_w1 is not None and cpu_w0 is not None else None
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's window before anything is torn down

    # ---- reduce over ranks ----
    ph_ms = torch.zeros(len(phases), dtype=torch.float64)
    for i in range(a.warmup, total_steps):
        for k in range(len(phases)):
            ph_ms[k] += K.elapsed_ms(ev[i][k], ev[i][k + 1])
    ph_ms /= a.steps
    t_el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    ph_max = ph_ms.to(dev).clone()
    errs_t = torch.tensor([memtest_errors], dtype=torch.int64, device=dev)
    gemm_flops = 2.0 * D * D * D * G
    hb = sz.hbm_bytes
    # per-rank rates; summed over ranks = the job's aggregate
    rates = torch.tensor([gemm_flops / (ph_ms[5].item() * 1e-3) / 1e12, gemm_flops / (ph_ms[6].item() * 1e-3) / 1e12,
                          hb / (ph_ms[0].item() * 1e-3) / 1e12, hb / (ph_ms[1].item() * 1e-3) / 1e12,
                          2 * hb / (ph_ms[2].item() * 1e-3) / 1e12, 3 * hb / (ph_ms[3].item() * 1e-3) / 1e12,
                          2 * hb / (ph_ms[4].item() * 1e-3) / 1e12], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t_el, op=dist.ReduceOp.MAX)
        dist.all_reduce(ph_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(errs_t, op=dist.ReduceOp.SUM)
        dist.all_reduce(rates, op=dist.ReduceOp.SUM)
    elapsed = t_el.item()
    ph_max = ph_max.cpu()
    rates = rates.cpu().tolist()
    ms_per_step = elapsed / a.steps * 1e3

    extras = {
        "mxfp8_tflops_aggregate": round(rates[1], 1),
        "hbm_read_tbps_per_gpu": round(rates[2] / world, 3),
        "hbm_write_tbps_per_gpu": round(rates[3] / world, 3),
        "hbm_copy_tbps_per_gpu": round(rates[4] / world, 3),
        "hbm_triad_tbps_per_gpu": round(rates[5] / world, 3),
        "memtest_tbps_per_gpu": round(rates[6] / world, 3),
        "memtest_errors": int(errs_t.item()),
        "bf16_tflops_per_gpu": round(rates[0] / world, 1),
        "mxfp8_tflops_per_gpu": round(rates[1] / world, 1),
        "phase_ms_max_over_ranks": {p: round(ph_max[k].item(), 3) for k, p in enumerate(phases)},
    }
    if world > 1:
        for k, op in ((7, "all_reduce"), (8, "all_gather"), (9, "all_to_all")):
            alg = sz.coll_bytes / (ph_max[k].item() * 1e-3) / 1e9
            extras[f"rccl_{op}_busbw_gbps"] = round(alg * busbw_factor(op, world), 3)
    if exporter is not None and scrape_ms:
        q = statistics.quantiles(scrape_ms, n=100) if len(scrape_ms) >= 2 else [scrape_ms[0]] * 99
        extras.update({
            "scrape_p50_ms": round(statistics.median(scrape_ms), 3),
            "scrape_p99_ms": round(q[98], 3),
            "scrape_bytes": scrape_bytes,
            "scrapes": len(scrape_ms),
            "exporter_cpu_percent_of_one_core": None if exporter_cpu_s is None
                                                else round(100.0 * exporter_cpu_s / max(exporter_window_s, 1e-9), 3),
            "exporter_cpu_window_s": round(exporter_window_s, 2),
            "exporter_rss_kb": exporter.rss_kb(),
            "exporter_interval_ms": 1000,
            "exporter_prof_counters": bool(a.prof_counters),
        })
        try:
            last = exporter.scrape()[1]
        except SCRAPE_ERRORS:
            last = b""
        m = re.search(rb'^dcgm_fi_prof_pipe_tensor_active\{gpu="0"[^}]*\} (\S+)', last, re.M)
        if m:
            extras["tensor_active_percent_gpu0_last_scrape"] = float(m.group(1))
        direct = re.findall(rb'^gpumon_engine_metrics_table_direct\{[^}]*\} (\d)', last, re.M)
        if direct:  # how the exporter reads the bulk metrics table: the driver's file itself, or through the device library
            extras["exporter_metrics_table_direct_gpus"] = sum(int(x) for x in direct)
    if scrape_failures:
        extras["scrape_error"] = scrape_failures[0]
        extras["exporter_exit_code"] = exporter.proc.poll()
    if extras["memtest_errors"]:
        extras["FAILED"] = "memtest reported errors"
    gib = lambda b: f"{b / 2**30:g}GiB"  # noqa: E731
    return {
        "metric": "diag_mfma_bf16_gemm_tflops",
        "value": round(rates[0], 1 if not a.dry_run else 6),
        "unit": "TFLOP/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": round(ms_per_step, 3),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "bf16",
        "data": "DRY RUN on CPU (toy sizes, not a measurement)" if a.dry_run
                else "synthetic (random normal operands, random-init buffers)",
        "config": {
            "model": f"gpumon diag suite: hbm stream {gib(sz.hbm_bytes)} + memtest {gib(sz.hbm_bytes)} + mfma gemm bf16/mxfp8 "
                     f"{D}^3 + rccl {gib(sz.coll_bytes)} + exporter scrape @1s",
            "global_batch": G * world,
            "seq_len": D,
            "parallelism": f"dp{world}",
        },
        "extras": extras,
    }


if __name__ == "__main__":
    main()
