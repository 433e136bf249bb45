#!/usr/bin/env python3
"""Scrape-latency and exporter-CPU benchmark: the operating point the exporter is
built for (8 GPUs, 1 s resolution, under 1 % of a host core, p50 scrape under 50 ms).

Starts `gpumon-exporter -e -p --extras --self-metrics --health --listen 127.0.0.1:0` as a
child (fake backend with N scripted GPUs, or amdsmi on a GPU box), scrapes
/metrics over a fresh HTTP connection at --scrape-hz for --seconds, and reports
latency percentiles, the child's on-CPU time as a share of wall time (all its
threads: sampler, HTTP, event waiter), RSS, body size and series count.

The reference's exporter is a bash+gawk pipeline with no published cost
(exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:85-193); its only figures
are the host engine's 0.08-0.165 % CPU and 11-18 MB RSS in sample outputs
(bindings/go/samples/dcgm/README.md:106-107, restApi/README.md:104).

  python scripts/scrape_bench.py --backend fake --gpus 8 --seconds 60
  python scripts/scrape_bench.py --backend amdsmi --seconds 30 --out gpurun_out/scrape_amdsmi.json
  python scripts/scrape_bench.py --backend amdsmi --replicate 8 --seconds 60 --interval-ms 100

--replicate N presents every real device N times (each read goes to the real
library once per replica), which prices an N-GPU node's sampling on a one-GPU box.
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(q * len(xs)))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", default="fake", choices=["fake", "amdsmi"])
    ap.add_argument("--gpus", type=int, default=8, help="fake backend only")
    ap.add_argument("--seconds", type=float, default=30)
    ap.add_argument("--scrape-hz", type=float, default=1.0)
    ap.add_argument("--replicate", type=int, default=0, help="present every device of the backend N times")
    ap.add_argument("--interval-ms", type=int, default=1000, help="exporter sampling interval (its floor is 100)")
    ap.add_argument("--health", action="store_true", help="kept for old command lines: the exporter child always runs the health check")
    ap.add_argument("--out")
    a = ap.parse_args()

    from bench import ExporterChild
    from k8s_gpu_monitor_amd.utils.promparse import parse
    spec = [f"gpus={a.gpus}", "seed=1"] if a.backend == "fake" else []
    if a.replicate > 1:
        spec.append(f"replicate={a.replicate}")
    if spec:
        os.environ["GPUMON_BACKEND_SPEC"] = ",".join(spec)
    ex = ExporterChild(a.backend, interval_ms=a.interval_ms, extra_args=[])  # the child always runs with --health
    try:
        time.sleep(1.5)  # first sweep done, static attributes cached
        ex.scrape()
        rss0 = ex.rss_kb()
        cpu0, t0 = ex.cpu_seconds(), time.perf_counter()
        lat, body = [], b""
        period = 1.0 / a.scrape_hz
        nxt = t0
        while time.perf_counter() - t0 < a.seconds:
            ms, body = ex.scrape()
            lat.append(ms)
            if len(lat) % max(1, int(30 * a.scrape_hz)) == 0:  # a long run says that it is alive
                print(f"scrape_bench: {time.perf_counter() - t0:.0f} s, {len(lat)} scrapes, RSS {ex.rss_kb()} kB", flush=True)
            nxt += period
            time.sleep(max(0.0, nxt - time.perf_counter()))
        wall = time.perf_counter() - t0
        cpu = ex.cpu_seconds() - cpu0
        fams = parse(body.decode())
        series = sum(len(f.samples) for f in fams.values())
        gpus = len({l.get("gpu") for l, _ in fams["dcgm_sm_clock"].samples})
        self_metrics = {n: f.samples[0][1] for n, f in fams.items() if n.startswith("gpumon_") and len(f.samples) == 1}
        res = {
            "backend": a.backend, "health": True, "replicate": max(1, a.replicate), "gpus": gpus, "sample_interval_ms": a.interval_ms, "scrape_hz": a.scrape_hz,
            "seconds": round(wall, 1), "scrapes": len(lat),
            "scrape_ms": {"p50": round(statistics.median(lat), 3), "p90": round(pct(lat, 0.9), 3),
                          "p99": round(pct(lat, 0.99), 3), "max": round(max(lat), 3)},
            "exporter_cpu_percent_of_one_core": round(100 * cpu / wall, 3),
            "exporter_rss_kb": ex.rss_kb(), "exporter_rss_kb_at_start": rss0, "body_bytes": len(body), "families": len(fams), "series": series,
            "self_metrics": self_metrics,
            "targets": {"scrape_p50_ms": 50, "cpu_percent": 1.0},
        }
        res["meets_targets"] = res["scrape_ms"]["p50"] < 50 and res["exporter_cpu_percent_of_one_core"] < 1.0
    finally:
        ex.stop()
    text = json.dumps(res, indent=1)
    print(text)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        with open(a.out, "w") as f:
            f.write(text + "\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
