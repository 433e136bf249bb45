"""bench.py's driver contract, exercised without GPUs: `--dry-run` runs the same
rank handling, collectives, reductions and JSON assembly on CPU tensors over gloo
(toy sizes), with the exporter child on the fake backend."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config"}


def _port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_single_rank_dry_run():
    r = subprocess.run([sys.executable, "bench.py", "--dry-run", "--steps", "6", "--warmup", "2"], cwd=ROOT, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "bf16" and "DRY RUN" in d["data"]
    assert d["config"]["parallelism"] == "dp1" and d["value"] > 0 and d["ms_per_step"] > 0
    x = d["extras"]
    assert x["scrapes"] == 6 and x["scrape_p50_ms"] < 50 and x["scrape_bytes"] > 5000  # 8 fake GPUs behind the exporter child
    assert "rccl_all_reduce_busbw_gbps" not in x and x["memtest_errors"] == 0


@pytest.mark.parametrize("n", [2, 4])
def test_multi_rank_dry_run_under_torchrun(n):
    """The launch line the driver uses, with gloo instead of RCCL."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr",
                        "127.0.0.1", "--master-port", str(_port()), "bench.py", "--gpus", str(n), "--steps", "5", "--warmup",
                        "1", "--dry-run"], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    assert d["n_gpus"] == n and d["config"]["parallelism"] == f"dp{n}" and d["config"]["global_batch"] == 8 * n
    x = d["extras"]
    assert d["value"] == pytest.approx(x["bf16_tflops_per_gpu"] * n, abs=0.05 * n + 1e-3)  # whole-job aggregate (per-GPU is rounded)
    for op in ("all_reduce", "all_gather", "all_to_all"):
        assert x[f"rccl_{op}_busbw_gbps"] > 0
        assert op in x["phase_ms_max_over_ranks"]
    assert x["scrapes"] == 5


def test_a_dying_exporter_does_not_take_the_measurement_with_it():
    """The monitoring child is killed before step 3 of 6: the kernels' numbers are still reported, the scrape phase ends at
    the first failure and the JSON says so."""
    r = subprocess.run([sys.executable, "bench.py", "--dry-run", "--steps", "6", "--warmup", "0", "--kill-exporter-at", "3"], cwd=ROOT,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    d = _json_line(r.stdout)
    x = d["extras"]
    assert d["value"] > 0 and x["scrapes"] == 3 and "step 3" in x["scrape_error"] and x["exporter_exit_code"] == -9


def test_scrape_benchmark_on_eight_scripted_gpus(tmp_path):
    """The 8-GPU / 1 s operating point: p50 scrape under 50 ms, exporter under 1 % of a core."""
    out = tmp_path / "scrape.json"
    r = subprocess.run([sys.executable, "scripts/scrape_bench.py", "--backend", "fake", "--gpus", "8", "--seconds", "4",
                        "--scrape-hz", "4", "--out", str(out)], cwd=ROOT, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(out.read_text())
    assert d["gpus"] == 8 and d["scrapes"] >= 12 and d["series"] > 500
    assert d["scrape_ms"]["p50"] < 50 and d["exporter_cpu_percent_of_one_core"] < 5  # CI machines are noisy; measured 0.15 %
    assert d["self_metrics"]["gpumon_engine_backend_errors_total"] == 0


def test_gpu_count_mismatch_is_refused():
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--dry-run"], cwd=ROOT, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "launcher" in r.stderr
