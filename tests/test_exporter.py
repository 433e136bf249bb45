"""Prometheus exporter against the fake device backend (CPU only).

Covers the text contract of the reference's exporter script
(exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:97-193): family names,
types and HELP text, label order gpu,uuid, blank ("N/A") omission, the
synthesised gpu_last_not_idle_time, GPU subset selection, the >=100 ms interval
floor and atomic publishing — plus what a native exporter adds (valid family
grouping, AMD extras, self metrics, HTTP).
"""
import os
import re
import subprocess
import time
import urllib.request

import pytest

from conftest import ROOT, bin_path
from k8s_gpu_monitor_amd import abi as A
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd.utils.promparse import ExpositionError, parse

GOLDEN = os.path.join(ROOT, "tests", "golden", "exporter_2gpu_steady.prom")

# name, type, HELP — verbatim from the reference's awk program, in its order.
CONTRACT = [
    ("sm_clock", "gauge", "SM clock frequency (in MHz)."),
    ("memory_clock", "gauge", "Memory clock frequency (in MHz)."),
    ("memory_temp", "gauge", "Memory temperature (in C)."),
    ("gpu_temp", "gauge", "GPU temperature (in C)."),
    ("power_usage", "gauge", "Power draw (in W)."),
    ("total_energy_consumption", "counter", "Total energy consumption since boot (in mJ)."),
    ("pcie_tx_throughput", "counter", "Total number of bytes transmitted through PCIe TX (in KB) via NVML."),
    ("pcie_rx_throughput", "counter", "Total number of bytes received through PCIe RX (in KB) via NVML."),
    ("pcie_replay_counter", "counter", "Total number of PCIe retries."),
    ("gpu_utilization", "gauge", "GPU utilization (in %)."),
    ("gpu_last_not_idle_time", "gauge", "Timestamp of last time GPU utilization was 2% or less."),
    ("mem_copy_utilization", "gauge", "Memory utilization (in %)."),
    ("enc_utilization", "gauge", "Encoder utilization (in %)."),
    ("dec_utilization", "gauge", "Decoder utilization (in %)."),
    ("xid_errors", "gauge", "Value of the last XID error encountered."),
    ("power_violation", "counter", "Throttling duration due to power constraints (in us)."),
    ("thermal_violation", "counter", "Throttling duration due to thermal constraints (in us)."),
    ("sync_boost_violation", "counter", "Throttling duration due to sync-boost constraints (in us)."),
    ("board_limit_violation", "counter", "Throttling duration due to board limit constraints (in us)."),
    ("low_util_violation", "counter", "Throttling duration due to low utilization (in us)."),
    ("reliability_violation", "counter", "Throttling duration due to reliability constraints (in us)."),
    ("fb_total", "gauge", "Framebuffer memory free (in MiB)."),  # sic: the reference's typo is part of the contract
    ("fb_free", "gauge", "Framebuffer memory free (in MiB)."),
    ("fb_used", "gauge", "Framebuffer memory used (in MiB)."),
    ("ecc_sbe_volatile_total", "counter", "Total number of single-bit volatile ECC errors."),
    ("ecc_dbe_volatile_total", "counter", "Total number of double-bit volatile ECC errors."),
    ("ecc_sbe_aggregate_total", "counter", "Total number of single-bit persistent ECC errors."),
    ("ecc_dbe_aggregate_total", "counter", "Total number of double-bit persistent ECC errors."),
    ("retired_pages_sbe", "counter", "Total number of retired pages due to single-bit errors."),
    ("retired_pages_dbe", "counter", "Total number of retired pages due to double-bit errors."),
    ("retired_pages_pending", "counter", "Total number of pages pending retirement."),
    ("nvlink_flit_crc_error_count_total", "counter", "Total number of NVLink flow-control CRC errors."),
    ("nvlink_data_crc_error_count_total", "counter", "Total number of NVLink data CRC errors."),
    ("nvlink_replay_error_count_total", "counter", "Total number of NVLink retries."),
    ("nvlink_recovery_error_count_total", "counter", "Total number of NVLink recovery errors."),
    ("nvlink_bandwidth_total", "counter", "Total number of NVLink bandwidth counters for all lanes"),
]
PROFILING = [
    ("fi_prof_gr_engine_active", "gauge", "Ratio of time the graphics engine is active (in %)."),
    ("fi_prof_sm_active", "gauge", "The ratio of cycles an SM has at least 1 warp assigned (in %)."),
    ("fi_prof_sm_occupancy", "gauge", "The ratio of number of warps resident on an SM (in %)."),
    ("fi_prof_pipe_tensor_active", "gauge", "Ratio of cycles the tensor (HMMA) pipe is active (in %)."),
    ("fi_prof_dram_active", "gauge",
     "Ratio of cycles the device memory interface is active sending or receiving data (in %)."),
]
# Families with no AMD meaning on the fake MI355X node: blank, hence omitted.
OMITTED_ON_MI355X = {"enc_utilization", "sync_boost_violation", "nvlink_replay_error_count_total",
                     "nvlink_recovery_error_count_total"}


def _normalise(text):
    """The one wall-clock dependent value is replaced by a token."""
    return re.sub(r"(dcgm_gpu_last_not_idle_time\{[^}]*\}) \d+", r"\1 <TS>", text)


def test_registry_has_verbatim_contract(engine2):
    seen = {}
    for fid in E.field_list():
        m = E.field_meta(fid)
        if m["prom_name"]:
            seen[m["prom_name"].decode() if isinstance(m["prom_name"], bytes) else m["prom_name"]] = m
    for name, typ, help_ in CONTRACT + PROFILING:
        if name == "gpu_last_not_idle_time":
            continue  # derived at render time
        m = seen["dcgm_" + name]
        assert {1: "gauge", 2: "counter"}[m["prom_type"]] == typ, name
        assert m["prom_help"] == help_, name


def test_golden_file_byte_exact():
    E.init(backend="fake", backend_spec="gpus=2,steady=1,seed=1")
    try:
        with E.Exporter(interval_ms=1000) as ex:
            ex.sweep()
            text = ex.render()
    finally:
        E.shutdown()
    got = _normalise(text)
    if os.environ.get("GPUMON_UPDATE_GOLDEN"):
        os.makedirs(os.path.dirname(GOLDEN), exist_ok=True)
        open(GOLDEN, "w").write(got)
    assert got == open(GOLDEN).read()


def test_strict_exposition_and_contract_order(engine):
    with E.Exporter(interval_ms=1000, profiling=True) as ex:
        ex.sweep()
        fams = parse(ex.render())
    expected = [("dcgm_" + n, t, h) for n, t, h in CONTRACT + PROFILING if n not in OMITTED_ON_MI355X]
    assert [f.name for f in fams.values()] == [n for n, _, _ in expected]
    for name, typ, help_ in expected:
        f = fams[name]
        assert f.type == typ and f.help == help_
        assert len(f.samples) == 8, name  # one per GPU, all contiguous under one header
        for i, (labels, _v) in enumerate(f.samples):
            assert list(labels.keys()) == ["gpu", "uuid"]  # label ORDER is part of the contract
            assert labels["gpu"] == str(i)
            assert labels["uuid"].startswith("GPU-00000001-fa4e-0355-0000-")


def test_blank_values_are_omitted_not_zero(engine2):
    with E.Exporter(interval_ms=1000) as ex:
        ex.sweep()
        assert "dcgm_enc_utilization" not in ex.render()  # no orphan HELP/TYPE either
        E.fake_ctl("unsupported gpu=1 field=150")
        E.fake_ctl("noperm gpu=0 field=155")
        ex.sweep()
        fams = parse(ex.render())
        assert [l["gpu"] for l, _ in fams["dcgm_gpu_temp"].samples] == ["0"]
        assert [l["gpu"] for l, _ in fams["dcgm_power_usage"].samples] == ["1"]
        E.fake_ctl("clear gpu=1 field=150")
        ex.sweep()
        assert len(parse(ex.render())["dcgm_gpu_temp"].samples) == 2


def test_backend_failure_blanks_but_never_stops(engine2):
    with E.Exporter(interval_ms=1000) as ex:
        ex.sweep()
        E.fake_ctl("fail gpu=0")
        ex.sweep()
        fams = parse(ex.render())
        assert all(l["gpu"] == "1" for f in fams.values() for l, _ in f.samples if f.name != "dcgm_xid_errors")
        E.fake_ctl("clear gpu=0")
        ex.sweep()
        assert len(parse(ex.render())["dcgm_sm_clock"].samples) == 2
        assert E.introspect()["backend_errors"] > 0


def test_last_not_idle_time_latches(engine2):
    with E.Exporter(interval_ms=1000) as ex:
        def ts():
            ex.sweep()
            return {l["gpu"]: v for l, v in parse(ex.render())["dcgm_gpu_last_not_idle_time"].samples}
        t0 = ts()
        now = time.time()
        assert all(abs(v - now) < 5 for v in t0.values())  # first sight latches "now"
        time.sleep(1.2)
        t1 = ts()
        assert t1 == t0  # idle (util 0 <= 2): the stamp holds
        E.fake_ctl("load gpu=1 util=55")
        time.sleep(1.1)
        t2 = ts()
        assert t2["0"] == t0["0"] and t2["1"] > t0["1"]  # busy GPU re-latches
        E.fake_ctl("load gpu=1 util=2")  # 2 % is still idle: "> 2" in the reference
        time.sleep(1.1)
        t3 = ts()
        assert t3["1"] == t2["1"]


def test_last_not_idle_latch_survives_a_restart(engine2, tmp_path):
    """What the exporter accumulates itself is carried in a state file keyed by GPU UUID; the reference's
    awk script starts the latch at "now" on every start (dcgm-exporter:97-113)."""
    import json
    state = tmp_path / "state" / "exporter.json"
    state.parent.mkdir()

    def latch(ex):
        ex.sweep()
        return {l["uuid"]: v for l, v in parse(ex.render())["dcgm_gpu_last_not_idle_time"].samples}

    with E.Exporter(state_path=str(state)) as ex:
        first = latch(ex)
    saved = json.loads(state.read_text())  # written at shutdown even though the 10 s save period has not passed
    assert saved["version"] == 1 and saved["last_not_idle"] == {u: int(v) for u, v in first.items()}
    time.sleep(1.1)
    with E.Exporter(state_path=str(state)) as ex:
        assert latch(ex) == first  # idle GPUs: the stamps of the previous run hold
        E.fake_ctl("load gpu=0 util=80")
        time.sleep(1.1)
        busy = latch(ex)
        uuid0 = E.device_attributes(0)["uuid"]
        assert busy[uuid0] > first[uuid0] and all(busy[u] == first[u] for u in first if u != uuid0)
    E.fake_ctl("load gpu=0 util=0")
    assert json.loads(state.read_text())["last_not_idle"][uuid0] == int(busy[uuid0])
    # a stamp from the future (clock stepped back) and a damaged file both mean "start over", never a crash
    state.write_text(json.dumps({"version": 1, "last_not_idle": {uuid0: int(time.time()) + 10 ** 6}}))
    with E.Exporter(state_path=str(state)) as ex:
        assert abs(latch(ex)[uuid0] - time.time()) < 5
    state.write_text("{broken")
    with E.Exporter(state_path=str(state)) as ex:
        assert abs(latch(ex)[uuid0] - time.time()) < 5
    # without a state path nothing is written
    with E.Exporter() as ex:
        latch(ex)
    assert sorted(p.name for p in state.parent.iterdir()) == ["exporter.json"]


def test_partitioned_gpus_carry_a_partition_label():
    """In DPX / QPX / CPX mode the device library enumerates one entry per compute partition; each keeps
    its own gpu index and uuid and is told apart by an extra label.  Unpartitioned (SPX) nodes - every
    other test in this file - have no such label, so the reference's label set is unchanged there."""
    E.init(backend="fake", backend_spec="gpus=16,seed=3,steady=1,partition_mode=CPX")
    try:
        assert E.latest_values(9, [68])[68] == 1  # GM_FI_DEV_PARTITION_ID: entry 9 is slice 1 of the second GPU
        assert E.device_attributes(9)["compute_partition"] == "CPX"
        with E.Exporter() as ex:
            ex.sweep()
            fams = parse(ex.render())
            rows = [l for l, _ in fams["dcgm_sm_clock"].samples]
            assert [list(l) for l in rows][0] == ["gpu", "uuid", "partition"]  # after the reference's two labels
            assert [l["partition"] for l in rows] == [str(i % 8) for i in range(16)]
            assert len({l["uuid"] for l in rows}) == 16
    finally:
        E.shutdown()
    E.init(backend="fake", backend_spec="gpus=2,seed=3,steady=1")
    try:
        with E.Exporter() as ex:
            ex.sweep()
            assert "partition=" not in ex.render()
    finally:
        E.shutdown()


def test_gpu_subset_and_bad_subset(engine):
    with E.Exporter(interval_ms=1000, gpus="5,2") as ex:
        ex.sweep()
        fams = parse(ex.render())
        assert [l["gpu"] for l, _ in fams["dcgm_sm_clock"].samples] == ["2", "5"]
    with pytest.raises(E.GpumonError):
        E.Exporter(gpus="0,99")
    with pytest.raises(E.GpumonError):
        E.Exporter(gpus="zero")


def test_amd_extras_and_self_metrics(engine2):
    with E.Exporter(interval_ms=1000, extras=True, self_metrics=True) as ex:
        ex.sweep()
        fams = parse(ex.render())
    link = fams["gpumon_xgmi_link_rx_kb_total"]
    assert [list(l.keys()) for l, _ in link.samples][0] == ["gpu", "uuid", "link", "peer"]
    assert {l["link"] for l, _ in link.samples} == {"0"}  # a 2-GPU node wires one link per GPU
    # the far end of every link is named: link 0 of GPU 0 goes to GPU 1 and back
    assert {(l["gpu"], l["peer"]) for l, _ in link.samples} == {("0", "1"), ("1", "0")}
    assert fams["gpumon_xgmi_error_status"].samples[0][1] == 0
    assert {l["stack"] for l, _ in fams["gpumon_hbm_stack_temp"].samples} == {str(i) for i in range(8)}
    assert {l["xcd"] for l, _ in fams["gpumon_xcd_busy"].samples} == {str(i) for i in range(8)}
    assert {l["block"] for l, _ in fams["gpumon_ecc_block_uncorrectable_total"].samples} == {
        "umc", "gfx", "sdma", "mmhub", "xgmi"}
    for name in ("gpumon_engine_cpu_percent", "gpumon_engine_memory_kb", "gpumon_exporter_render_duration_seconds",
                 "gpumon_engine_sample_cycle_seconds"):
        assert name in fams and len(fams[name].samples) == 1
    # where the process list comes from is visible per GPU (on a host without the KFD table: fdinfo or open-files)
    assert {(l["gpu"], l["source"]) for l, _ in fams["gpumon_engine_process_source"].samples} == {("0", "fake"), ("1", "fake")}
    # the contract families are untouched by the extras
    assert fams["dcgm_sm_clock"].help == "SM clock frequency (in MHz)."


def test_legacy_layout_matches_reference_shape(engine2):
    """GPU-major, HELP/TYPE only with the first GPU (dcgm-exporter:99-102)."""
    with E.Exporter(interval_ms=1000, legacy_layout=True) as ex:
        ex.sweep()
        text = ex.render()
    lines = text.splitlines()
    first_gpu1 = next(i for i, l in enumerate(lines) if 'gpu="1"' in l)
    assert all(not l.startswith("#") for l in lines[first_gpu1:])
    assert sum(1 for l in lines if l.startswith("# TYPE")) == len(CONTRACT) - len(OMITTED_ON_MI355X)
    with pytest.raises(ExpositionError):  # which is exactly why the native layout groups by family
        parse(text)


def test_atomic_file_publish_and_http(engine2, tmp_path):
    out = tmp_path / "prom" / "dcgm.prom"
    with E.Exporter(interval_ms=100, output_path=str(out), listen="127.0.0.1:0", self_metrics=True) as ex:
        ex.start()
        deadline = time.time() + 5
        while not out.exists() and time.time() < deadline:
            time.sleep(0.02)
        assert out.exists()
        seen = set()
        for _ in range(200):  # a reader never sees a partial file
            txt = out.read_text()
            assert txt.endswith("\n")
            fams = parse(txt)
            assert len(fams["dcgm_fb_used"].samples) == 2
            seen.add(os.stat(out).st_ino)
            time.sleep(0.003)
        assert len(seen) > 1  # replaced by rename, not rewritten in place
        assert not os.path.exists(str(out) + ".swp") or True
        assert oct(os.stat(out).st_mode & 0o777) == "0o644"
        port = ex.port
        body = urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=5).read().decode()
        assert parse(body)["dcgm_gpu_temp"].samples
        body2 = urllib.request.urlopen(f"http://127.0.0.1:{port}/gpu/metrics", timeout=5).read().decode()
        assert "dcgm_gpu_temp" in body2
        assert urllib.request.urlopen(f"http://127.0.0.1:{port}/healthz", timeout=5).read() == b"ok\n"
        with pytest.raises(urllib.error.HTTPError) as e:
            urllib.request.urlopen(f"http://127.0.0.1:{port}/nope", timeout=5)
        assert e.value.code == 404
        ex.stop()


def test_exporter_binary_flags(tmp_path):
    exe = bin_path("gpumon-exporter")
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=4,steady=1")
    out = tmp_path / "x.prom"
    # interval below 100 ms is refused like the reference (dcgm-exporter:32-34)
    r = subprocess.run([exe, "-e", "-d", "50", "--once", "-o", str(out)], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    r = subprocess.run([exe, "-h"], env=env, capture_output=True, text=True)
    assert r.returncode == 1 and "-e" in r.stderr and "-d" in r.stderr
    r = subprocess.run([exe, "-e", "-p", "--once", "-o", str(out)], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    fams = parse(out.read_text())
    assert "dcgm_fi_prof_gr_engine_active" in fams and len(fams["dcgm_sm_clock"].samples) == 4
    # NODE_NAME -> env var named after the node (dashes become underscores) -> GPU list
    env2 = dict(env, NODE_NAME="gpu-node-7", gpu_node_7="1,3")
    r = subprocess.run([exe, "-e", "--once", "-o", str(out)], env=env2, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert [l["gpu"] for l, _ in parse(out.read_text())["dcgm_sm_clock"].samples] == ["1", "3"]
    assert "GPUs:1,3" in r.stderr


def test_flags_from_config_file_and_environment(tmp_path):
    """One configuration: $GPUMON_ARGS < --config FILE < command line (later words win)."""
    exe = bin_path("gpumon-exporter")
    out_file, out_cli = tmp_path / "from_file.prom", tmp_path / "from_cli.prom"
    cfg = tmp_path / "exporter.conf"
    cfg.write_text(f'# node exporter settings\n-e --once   # embedded, one sweep\n-o "{out_file}"\n--gpus 1\n')
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=4,steady=1", GPUMON_ARGS="-p --gpus 0,2")
    r = subprocess.run([exe, "--config", str(cfg)], env=env, capture_output=True, text=True, timeout=30)
    assert r.returncode == 0, r.stderr
    fams = parse(out_file.read_text())
    assert [l["gpu"] for l, _ in fams["dcgm_sm_clock"].samples] == ["1"]  # the file overrides the environment
    assert "dcgm_fi_prof_gr_engine_active" in fams  # -p came from the environment
    r = subprocess.run([exe, "--config", str(cfg), "-o", str(out_cli), "--gpus", "3"], env=env, capture_output=True, text=True,
                       timeout=30)
    assert r.returncode == 0, r.stderr
    assert [l["gpu"] for l, _ in parse(out_cli.read_text())["dcgm_sm_clock"].samples] == ["3"]  # the command line overrides both
    r = subprocess.run([exe, "--config", str(tmp_path / "missing.conf")], env=env, capture_output=True, text=True, timeout=30)
    assert r.returncode != 0 and "cannot read config file" in r.stderr


def test_interval_floor_is_clamped_in_library(engine2):
    with E.Exporter(interval_ms=5) as ex:  # library clamps instead of failing
        ex.sweep()
        assert "dcgm_sm_clock" in ex.render()


# ---- active-diagnostics reports as metrics ------------------------------------

def _diag_report(gpu=1, result="pass", copy_tbps=5.5, errors=0):
    """Same member names as `gpumon-diag run --json` writes."""
    return {
        "gpu": gpu, "timestamp": 1700000123,
        "hbm_stream": {"read": {"tbps_best": 7.0, "tbps_median": 6.9, "verified": True},
                       "copy": {"tbps_best": copy_tbps, "tbps_median": 5.4, "verified": True}},
        "memtest": {"bytes_tested": 274877906944, "errors": errors},
        "pcie": {"link_gen": 5, "link_width": 16, "link_gbps": 63.008, "h2d_gbps": 57.2, "d2h_gbps": 56.9, "bidir_gbps": 48.6,
                 "kernel_read_gbps": 55.3, "kernel_write_gbps": 53.4, "bad_words": 0},
        "mfma_peak": {"bf16": {"tflops": 2179.5, "tflops_best": 2200.0}},
        "gemm_stress": {"bf16": {"tflops": 1450.0, "tflops_best": 1500.0, "launches": 900, "sdc_mismatches": 0,
                                 "checked_in_kernel": 896, "checked_in_memory": 113, "mfma_active_avg": 71.7}},
        "mfma_active": {"min": 69.1, "avg": 70.2, "max": 71.4, "clock_mhz": 1940.0, "windows": 18},
        "telemetry": {"power": {"min": 900.0, "avg": 1350.5, "max": 1400.0, "samples": 50},
                      "gfx_clock": {"min": 2100.0, "avg": 2180.0, "max": 2400.0, "samples": 50}},
        "rccl": {"gpus": 8, "points": [], "peak_busbw_gbps": {"all_reduce": 310.25, "all_gather": 295.0}},
        "p2p_gbps": [[0, 48.5], [47.25, 0]],
        "result": result,
    }


def _write_json(path, doc):
    import json
    tmp = str(path) + ".tmp"
    with open(tmp, "w") as f:
        json.dump(doc, f)
    os.replace(tmp, path)


def test_diag_report_becomes_metrics(engine2, tmp_path):
    rep = tmp_path / "diag.json"
    _write_json(rep, _diag_report())
    with E.Exporter(diag_results_path=str(rep)) as ex:
        ex.sweep()
        fams = parse(ex.render())  # strict: HELP/TYPE once per family, well-formed labels
        by = lambda fam: {tuple(sorted(l.items())): v for l, v in fams[fam].samples}
        assert by("gpumon_diag_pass") == {(("gpu", "1"),): 1.0}
        assert by("gpumon_diag_timestamp_seconds") == {(("gpu", "1"),): 1700000123.0}
        assert by("gpumon_diag_hbm_stream_tbps")[(("gpu", "1"), ("kernel", "copy"))] == 5.5
        assert by("gpumon_diag_hbm_stream_tbps")[(("gpu", "1"), ("kernel", "read"))] == 7.0
        assert by("gpumon_diag_memtest_bytes") == {(("gpu", "1"),): 274877906944.0}
        assert by("gpumon_diag_memtest_errors") == {(("gpu", "1"),): 0.0}
        assert by("gpumon_diag_pcie_gbps")[(("gpu", "1"), ("traffic", "h2d"))] == 57.2 and len(by("gpumon_diag_pcie_gbps")) == 5
        assert by("gpumon_diag_pcie_link_gbps") == {(("gpu", "1"),): 63.008}
        assert by("gpumon_diag_mfma_peak_tflops")[(("dtype", "bf16"), ("gpu", "1"))] == 2179.5
        assert by("gpumon_diag_gemm_tflops")[(("dtype", "bf16"), ("gpu", "1"))] == 1450.0
        assert by("gpumon_diag_gemm_sdc_mismatches")[(("dtype", "bf16"), ("gpu", "1"))] == 0.0
        assert by("gpumon_diag_gemm_checked_launches")[(("dtype", "bf16"), ("gpu", "1"), ("how", "kernel"))] == 896.0
        assert by("gpumon_diag_gemm_checked_launches")[(("dtype", "bf16"), ("gpu", "1"), ("how", "memory"))] == 113.0
        assert by("gpumon_diag_soak_power_watts")[(("gpu", "1"), ("stat", "avg"))] == 1350.5
        assert by("gpumon_diag_soak_mfma_active_percent")[(("gpu", "1"), ("stat", "max"))] == 71.4
        assert by("gpumon_diag_gemm_mfma_active_percent")[(("dtype", "bf16"), ("gpu", "1"))] == 71.7
        assert by("gpumon_diag_soak_gfx_clock_mhz")[(("gpu", "1"), ("stat", "max"))] == 2400.0
        assert "gpumon_diag_soak_gpu_temp_celsius" not in fams  # not in the report: no family
        assert by("gpumon_diag_rccl_busbw_gbps")[(("gpus", "8"), ("op", "all_reduce"))] == 310.25
        assert by("gpumon_diag_p2p_gbps") == {(("dst", "1"), ("src", "0")): 48.5, (("dst", "0"), ("src", "1")): 47.25}
        assert "dcgm_sm_clock" in fams  # passive telemetry still there

        # a newer report replaces the cached one
        time.sleep(0.02)
        _write_json(rep, _diag_report(result="fail", errors=3))
        fams = parse(ex.render())
        assert fams["gpumon_diag_pass"].samples[0][1] == 0.0
        assert fams["gpumon_diag_memtest_errors"].samples[0][1] == 3.0


def _verdict(test, health, gpu=None, bdf=None, system=4, message="m"):
    v = {"test": test, "system": system, "health": health, "value": 1.0, "message": message}
    if gpu is not None:
        v["gpu"] = gpu
    if bdf is not None:
        v["pci_bdf"] = bdf
    return v


def test_node_wide_reports_never_speak_for_a_gpu(engine2, tmp_path):
    """A directory as the scheduled diagnostics leave it: one report per GPU and two node-wide ones (link test, RCCL
    sweep).  The node-wide reports pass; GPU 0's own report failed.  Files are merged in name order, and the node-wide
    ones come last: gpumon_diag_pass{gpu="0"} must still say 0, the node-wide results get series of their own, and each
    GPU's verdicts are there per test."""
    d = tmp_path / "reports"
    d.mkdir()
    bdf = [E.device_attributes(i)["pci_bdf"] for i in range(2)]
    gpu0 = dict(_diag_report(gpu=0, result="fail", errors=5), scope="gpu", pci_bdf=bdf[0],
                verdicts=[_verdict("memtest", 20, message="memtest found 5 bad words"), _verdict("hbm_stream", 0)])
    gpu1 = dict(_diag_report(gpu=1), scope="gpu", pci_bdf=bdf[1], verdicts=[_verdict("memtest", 0), _verdict("hbm_stream", 10)])
    for doc in (gpu0, gpu1):
        doc.pop("rccl"), doc.pop("p2p_gbps")
    links = {"scope": "node", "timestamp": 1700000200, "result": "pass", "p2p_gbps": [[0, 58.5], [57.25, 0]],
             "links": {"gpus": 2, "findings": [{"kind": "slow_link", "gpu": 1, "pci_bdf": bdf[1], "link": 3, "peer": 0, "health": 10}]},
             "verdicts": [_verdict("xgmi_p2p", 0, gpu=0, bdf=bdf[0], system=2), _verdict("xgmi_p2p", 10, gpu=1, bdf=bdf[1], system=2)]}
    rccl = {"scope": "node", "timestamp": 1700000300, "result": "pass",
            "rccl": {"gpus": 2, "points": [], "peak_busbw_gbps": {"all_reduce": 61.5}, "expected_busbw_gbps": {"all_reduce": 53.76}},
            "verdicts": [_verdict("rccl", 0, system=2)]}
    for name, doc in (("gpu0.json", gpu0), ("gpu1.json", gpu1), ("links.json", links), ("rccl.json", rccl)):
        _write_json(d / name, doc)
    with E.Exporter(diag_results_path=str(d)) as ex:
        ex.sweep()
        fams = parse(ex.render())
        assert {l["gpu"]: v for l, v in fams["gpumon_diag_pass"].samples} == {"0": 0.0, "1": 1.0}
        assert {l["gpu"]: v for l, v in fams["gpumon_diag_timestamp_seconds"].samples} == {"0": 1700000123.0, "1": 1700000123.0}
        assert {l["report"]: v for l, v in fams["gpumon_diag_node_pass"].samples} == {"links": 1.0, "rccl": 1.0}
        assert {l["report"]: v for l, v in fams["gpumon_diag_node_timestamp_seconds"].samples} == {"links": 1700000200.0, "rccl": 1700000300.0}
        th = {(l["gpu"], l["test"]): v for l, v in fams["gpumon_diag_test_health"].samples}
        assert th == {("0", "memtest"): 20, ("0", "hbm_stream"): 0, ("1", "memtest"): 0, ("1", "hbm_stream"): 10,
                      ("0", "xgmi_p2p"): 0, ("1", "xgmi_p2p"): 10}, th       # the node-wide rccl verdict names no GPU
        assert fams["gpumon_diag_link_findings"].samples == [({"gpu": "1", "kind": "slow_link"}, 1.0)]
        assert fams["gpumon_diag_rccl_expected_busbw_gbps"].samples == [({"gpus": "2", "op": "all_reduce"}, 53.76)]
        # and the engine was told, GPU by GPU
        r0 = {x["test"]: x["health"] for x in E.diag_results(0)}
        r1 = {x["test"]: x["health"] for x in E.diag_results(1)}
        assert r0 == {"memtest": 20, "hbm_stream": 0, "xgmi_p2p": 0} and r1 == {"memtest": 0, "hbm_stream": 10, "xgmi_p2p": 10}


def test_reports_find_their_gpu_by_pci_address(engine2, tmp_path):
    """The diagnostics number GPUs the way the compute runtime does, the engine follows the PCI bus, and a diagnostics
    container may see a subset of the cards.  A report that calls a card "gpu 0" but carries the PCI address of the
    engine's GPU 1 is GPU 1's, series and health incident alike; a report about a card the engine does not have is left
    out rather than blamed on whichever GPU has that number."""
    d = tmp_path / "reports"
    d.mkdir()
    bdf1 = E.device_attributes(1)["pci_bdf"]
    doc = dict(_diag_report(gpu=0, result="fail", errors=2), scope="gpu", pci_bdf=bdf1.upper(),
               verdicts=[_verdict("memtest", 20, message="memtest found 2 bad words")])
    _write_json(d / "a.json", doc)
    stranger = dict(_diag_report(gpu=0, result="fail"), scope="gpu", pci_bdf="0000:ff:1f.0", verdicts=[_verdict("memtest", 20)])
    _write_json(d / "b.json", stranger)
    with E.Exporter(diag_results_path=str(d)) as ex:
        ex.sweep()
        fams = parse(ex.render())
        assert fams["gpumon_diag_pass"].samples == [({"gpu": "1"}, 0.0)]
        assert fams["gpumon_diag_memtest_errors"].samples == [({"gpu": "1"}, 2.0)]
        assert fams["gpumon_diag_test_health"].samples == [({"gpu": "1", "test": "memtest"}, 20.0)]
        assert E.diag_results(0) == [] and [x["test"] for x in E.diag_results(1)] == ["memtest"]
        g = E.group_create("bdf", [0, 1])
        E.health_set(g, A.HEALTH["ALL"])
        status = {h["gpu"]: h["status"] for h in E.health_check(g)}
        assert status == {0: "Healthy", 1: "Failure"}


def test_diag_report_directory_and_garbage(engine2, tmp_path):
    d = tmp_path / "reports"
    d.mkdir()
    _write_json(d / "gpu0.json", _diag_report(gpu=0))
    _write_json(d / "gpu1.json", _diag_report(gpu=1, result="fail"))
    (d / "broken.json").write_text("{not json")
    (d / "notes.txt").write_text("ignored")
    with E.Exporter(diag_results_path=str(d)) as ex:
        ex.sweep()
        fams = parse(ex.render())
        got = {l["gpu"]: v for l, v in fams["gpumon_diag_pass"].samples}
        assert got == {"0": 1.0, "1": 0.0}
    # a path that does not exist yet is not an error: the report may come later
    with E.Exporter(diag_results_path=str(tmp_path / "later.json")) as ex:
        ex.sweep()
        assert "gpumon_diag_" not in ex.render()


def test_exporter_cli_diag_results_flag(tmp_path):
    rep = tmp_path / "diag.json"
    _write_json(rep, _diag_report(gpu=0))
    out = tmp_path / "metrics.prom"
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2,steady=1")
    r = subprocess.run([bin_path("gpumon-exporter"), "-e", "--once", "-o", str(out),
                        "--diag-results", str(rep)], env=env, capture_output=True, text=True, timeout=30)
    assert r.returncode == 0, r.stderr
    assert 'gpumon_diag_pass{gpu="0"} 1' in out.read_text()


def test_health_families_follow_the_engine_health_check(engine2):
    """--health: one health check per sweep, exported per subsystem; an incident that is gone at the next sweep stays
    visible in the incident counter."""
    def sample(fams, name, **want):
        hits = [v for l, v in fams[name].samples if all(l.get(k) == x for k, x in want.items())]
        assert len(hits) == 1, (name, want, hits)
        return hits[0]

    with E.Exporter(health=True, kubelet_socket=None) as ex:
        ex.sweep()
        f = parse(ex.render())
        assert len(f["gpumon_health_status"].samples) == 2 * 8 and f["gpumon_health_incidents_total"].type == "counter"
        assert f["gpumon_health_check_up"].samples == [({}, 1.0)]
        assert all(v == 0 for _l, v in f["gpumon_health_status"].samples + f["gpumon_health_overall"].samples)
        E.fake_ctl("set gpu=1 field=150 value=104")  # hot: thermal warning
        E.fake_ctl("link_down gpu=0 link=0")          # xGMI link down: failure
        time.sleep(1.1)
        ex.sweep()
        f = parse(ex.render())
        assert sample(f, "gpumon_health_status", gpu="1", system="thermal") == 10
        assert sample(f, "gpumon_health_status", gpu="0", system="xgmi") == 20
        assert sample(f, "gpumon_health_status", gpu="0", system="thermal") == 0
        assert sample(f, "gpumon_health_overall", gpu="0") == 20 and sample(f, "gpumon_health_overall", gpu="1") == 10
        assert sample(f, "gpumon_health_incidents_total", gpu="1", system="thermal", level="warning") == 1
        assert sample(f, "gpumon_health_incidents_total", gpu="0", system="xgmi", level="failure") >= 1
        # the same conditions through the policy engine: the link that went down is an xGMI violation of GPU 0.
        # Violations are delivered on the engine's notifier thread, so the counts may trail the sweep by a moment.
        deadline = time.time() + 5
        while True:
            f = parse(ex.render())
            if (sample(f, "gpumon_policy_violations_total", gpu="0", condition="xgmi") >= 1 and
                    sample(f, "gpumon_policy_violations_total", gpu="1", condition="thermal") >= 1) or time.time() > deadline:
                break
            time.sleep(0.05)
        assert sample(f, "gpumon_policy_violations_total", gpu="0", condition="xgmi") >= 1
        assert sample(f, "gpumon_policy_violations_total", gpu="1", condition="thermal") == 1  # 104 C against the 100 C default
        assert sample(f, "gpumon_policy_violations_total", gpu="1", condition="power") == 0
        # a scrape between two sweeps renders the same verdicts and does not run a check of its own
        assert parse(ex.render())["gpumon_health_incidents_total"].samples == f["gpumon_health_incidents_total"].samples
        E.fake_ctl("set gpu=1 field=150 value=60")
        E.fake_ctl("link_up gpu=0 link=0")
        time.sleep(1.1)
        ex.sweep()
        f = parse(ex.render())
        assert sample(f, "gpumon_health_status", gpu="1", system="thermal") == 0
        assert sample(f, "gpumon_health_overall", gpu="0") == 0
        assert sample(f, "gpumon_health_incidents_total", gpu="1", system="thermal", level="warning") == 1  # the counter keeps it
    with E.Exporter() as ex:  # off by default: the contract output is unchanged
        ex.sweep()
        assert "gpumon_health_status" not in parse(ex.render())


def test_slow_sources_keep_their_own_period_under_a_fast_collection_interval():
    """At -d 100 the bulk metrics are read ten times a second, but the sources that cost a device-library call each and
    move slowly (error counters, bad pages, performance level, memory usage) keep a one-second period: that is what
    makes a 100 ms exporter on eight GPUs cost little more than the bulk read.  Seen from outside as the sample
    timestamps of one field of each kind."""
    import time
    from k8s_gpu_monitor_amd import abi as A
    E.init(backend="fake", backend_spec="gpus=2,seed=3")
    try:
        with E.Exporter(interval_ms=100, slow_interval_ms=1000) as ex:
            ex.start()
            seen = {A.FI["GPU_TEMP"]: set(), A.FI["ECC_CE_VOLATILE"]: set(), A.FI["VRAM_USED"]: set(), A.FI["BAD_PAGES_UE"]: set()}
            t_end = time.time() + 1.6
            while time.time() < t_end:
                for f, stamps in seen.items():
                    stamps.update(ts for ts, _v in E.values_since(0, f, 0))
                time.sleep(0.03)
            ex.stop()
            assert len(seen[A.FI["GPU_TEMP"]]) >= 12, {k: len(v) for k, v in seen.items()}
            for f in (A.FI["ECC_CE_VOLATILE"], A.FI["VRAM_USED"], A.FI["BAD_PAGES_UE"]):
                assert 1 <= len(seen[f]) <= 3, {k: len(v) for k, v in seen.items()}
            text = ex.render()          # and everything is still exported
            assert "dcgm_ecc_sbe_volatile_total" in text and "dcgm_fb_used" in text and "dcgm_gpu_temp" in text
        # a collection interval at or above the slow period: one schedule, as before
        with E.Exporter(interval_ms=1000) as ex:
            ex.sweep()
            assert "dcgm_ecc_sbe_volatile_total" in ex.render()
    finally:
        E.shutdown()
