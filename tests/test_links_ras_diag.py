"""Per-link xGMI identity, RAS events derived from counter steps, throttle reasons / residency durations, health
thresholds as configuration, and active-diagnostics verdicts inside the engine — all on the scripted backend.
Reference behaviour: NVLink walk bindings/go/nvml/bindings.go:268-318 and nvml.go:539-568, XID events
bindings.go:68-146 and bindings/go/dcgm/policy.go:162-249, throttle reasons bindings.go:584-620."""
import json
import os
import subprocess
import time

import pytest

from conftest import bin_path
from k8s_gpu_monitor_amd import abi as A  # noqa: E402
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd import smi
from k8s_gpu_monitor_amd.utils.promparse import parse

F, H, P = A.FI, A.HEALTH, A.POLICY


# ------------------------------------------------------------------ xGMI links --
def test_every_link_names_its_far_end(engine):
    links = E.xgmi_links(3)
    assert [l["link"] for l in links] == list(range(7))  # fully connected hive of eight: seven links
    assert [l["peer_gpu"] for l in links] == [0, 1, 2, 4, 5, 6, 7]
    attrs = {g: E.device_attributes(g)["pci_bdf"] for g in range(8)}
    assert all(l["peer_bdf"] == attrs[l["peer_gpu"]] for l in links)
    assert all(l["state"] == 1 and l["bit_rate_gbps"] == 38 and l["max_bandwidth_gbps"] == 608 for l in links)
    # the link table and the per-link telemetry fields share their index
    E.fake_ctl("link_down gpu=3 link=4")
    E.update_all_fields(True)
    assert [l["state"] for l in E.xgmi_links(3)] == [1, 1, 1, 1, 0, 1, 1]
    # symmetric: the link of GPU 5 that goes to GPU 3 is the one whose peer is 3
    back = [l for l in E.xgmi_links(5) if l["peer_gpu"] == 3]
    assert len(back) == 1 and back[0]["link"] == 3
    # links per peer, the count the reference derives from the NVLink walk
    topo = E.device_topology(3)
    assert {p["gpu"]: p["num_links"] for p in topo["peers"]} == {g: 1 for g in range(8) if g != 3}
    with pytest.raises(E.GpumonError):
        E.xgmi_links(99)


def test_link_table_through_the_direct_layer_and_unsupported_platform():
    smi.init(backend="fake", backend_spec="gpus=4,seed=2")
    try:
        links = smi.xgmi_links(0)
        assert [(l["link"], l["peer_gpu"]) for l in links] == [(0, 1), (1, 2), (2, 3)]
        smi.fake_ctl("topology_unsupported on=1")
        with pytest.raises(smi.GpumonError) as e:
            smi.xgmi_links(0)
        assert e.value.status == A.GM_ST_NOT_SUPPORTED
    finally:
        smi.shutdown()


def test_exporter_labels_links_with_their_peer(engine):
    with E.Exporter(interval_ms=1000, extras=True) as ex:
        ex.sweep()
        fams = parse(ex.render())
    up = {(l["gpu"], l["link"]): l["peer"] for l, _ in fams["gpumon_xgmi_link_up"].samples}
    assert up[("0", "0")] == "1" and up[("7", "6")] == "6" and up[("3", "3")] == "4"
    assert len(up) == 8 * 7


# ------------------------------------------------------------------ RAS events --
def test_uncorrectable_step_is_one_event_one_violation_one_health_failure(engine):
    g = E.group_create("ras", [2])
    E.health_set(g, H["DRIVER"])
    E.policy_set(g, P["GPU_EVENT"])
    E.update_all_fields(True)
    E.health_check(g)  # baseline
    with E.EventSet(2) as es, E.PolicyRegistration(g, P["GPU_EVENT"]) as reg:
        E.update_all_fields(True)
        assert es.wait(200) is None and reg.wait(100) is None  # a steady counter is not news
        E.fake_ctl("ecc gpu=2 block=0 ue=3")
        E.update_all_fields(True)
        ev = es.wait(2000)
        assert ev and ev["code"] == A.EVENT["RAS_UE"] and ev["gpu"] == 2 and "3 uncorrectable" in ev["message"]
        v = reg.wait(2000)
        assert v and v["data"]["code"] == A.EVENT["RAS_UE"]
        E.update_all_fields(True)
        E.update_all_fields(True)
        assert es.wait(300) is None and reg.wait(100) is None  # exactly one
    assert E.latest_values(2, [F["LAST_EVENT_CODE"]])[F["LAST_EVENT_CODE"]] == A.EVENT["RAS_UE"]  # dcgm_xid_errors
    r = E.health_check(g)[0]
    assert r["status"] == "Failure" and [w["type"] for w in r["watches"]] == ["Driver / RAS event watches"]
    assert E.health_check(g)[0]["status"] == "Healthy"  # reported once


def test_deferred_error_step_is_a_poison_event(engine):
    with E.EventSet(0) as es:
        E.fake_ctl("ecc gpu=0 block=0 deferred=1")
        E.update_all_fields(True)  # first ECC sample: baseline or step, depending on whether one was taken before
        E.fake_ctl("ecc gpu=0 block=0 deferred=2")
        E.update_all_fields(True)
        codes = []
        while True:
            ev = es.wait(500)
            if not ev:
                break
            codes.append(ev["code"])
    assert codes and set(codes) == {A.EVENT["RAS_POISON"]}


# --------------------------------------------------- throttle reasons, durations --
def test_idle_gpu_low_utilisation_duration_never_outruns_the_wall_clock(engine):
    ids = [F["LOW_UTIL_VIOLATION"], F["THROTTLE_STATUS"]]
    fg = E.field_group_create("viol", ids)
    g = E.group_create("viol", [0, 1])
    E.watch_fields(g, fg, update_freq_us=50_000)
    E.fake_ctl("load gpu=1 util=80")
    E.update_all_fields(True)
    t0, v0 = time.time(), E.latest_values(0, ids)[F["LOW_UTIL_VIOLATION"]]
    time.sleep(0.6)
    E.update_all_fields(True)
    dt_us, v1 = (time.time() - t0) * 1e6, E.latest_values(0, ids)
    grown = v1[F["LOW_UTIL_VIOLATION"]] - v0
    # eight idle XCCs each accrue the whole interval; the device's duration is one interval, not eight
    assert 0.5 * dt_us < grown <= 1.02 * dt_us, (grown, dt_us)
    assert v1[F["THROTTLE_STATUS"]] & A.THROTTLE["LOW_UTIL"]
    busy = E.latest_values(1, ids)
    assert not busy[F["THROTTLE_STATUS"]] & A.THROTTLE["LOW_UTIL"]
    E.fake_ctl("violation gpu=1 reliability=500 power=100")
    E.update_all_fields(True)
    bits = E.latest_values(1, ids)[F["THROTTLE_STATUS"]]
    assert bits & A.THROTTLE["HOST_LIMIT"] and bits & A.THROTTLE["POWER"] and not bits & A.THROTTLE["THERMAL"]
    E.update_all_fields(True)
    assert E.latest_values(1, ids)[F["THROTTLE_STATUS"]] == 0  # reasons are "since the previous sample"


# ------------------------------------------------------------ health thresholds --
def test_health_thresholds_are_configuration(engine):
    d = E.health_thresholds_get()
    assert (d["hotspot_warn_c"], d["hotspot_fail_c"], d["mem_warn_c"], d["mem_fail_c"], d["bad_pages_fail"], d["ce_burst_warn"]) == (
        100, 110, 95, 105, 256, 100)
    g = E.group_create("th", [0])
    E.health_set(g, H["THERMAL"] | H["MEM"])
    E.fake_ctl("set gpu=0 field=150 value=92")
    E.update_all_fields(True)
    assert E.health_check(g)[0]["status"] == "Healthy"
    E.health_thresholds_set(hotspot_warn_c=90, hotspot_fail_c=95, ce_burst_warn=5)
    assert E.health_check(g)[0]["status"] == "Warning"
    E.fake_ctl("set gpu=0 field=150 value=96")
    E.fake_ctl("ecc gpu=0 block=0 ce=6")
    E.update_all_fields(True)
    r = E.health_check(g)[0]
    assert r["status"] == "Failure" and sorted(w["type"] for w in r["watches"]) == ["Memory watches", "Thermal watches"]
    with pytest.raises(E.GpumonError):
        E.health_thresholds_set(hotspot_warn_c=120, hotspot_fail_c=100)  # fail below warn
    with pytest.raises(ValueError):
        E.health_thresholds_set(no_such_level=1)


def test_daemons_take_health_thresholds_as_flags(tmp_path):
    sock = str(tmp_path / "he.sock")
    p = subprocess.Popen([bin_path("gpumon-hostengine"), "--domain-socket", sock, "--backend", "fake", "--backend-spec", "gpus=1",
                          "--health-hotspot-warn", "70", "--health-hotspot-fail", "80"], stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        E.init(mode=E.STANDALONE, address=sock, is_unix_socket=True)
        try:
            d = E.health_thresholds_get()
            assert (d["hotspot_warn_c"], d["hotspot_fail_c"], d["mem_warn_c"]) == (70, 80, 95)
        finally:
            E.shutdown()
    finally:
        p.terminate()
        p.wait(timeout=10)


# ------------------------------------------------------- diagnostics -> engine --
def test_failed_diagnostic_is_an_incident_an_event_and_a_violation(engine):
    g = E.group_create("diag", [1])
    E.health_set(g, H["ALL"])
    E.policy_set(g, P["GPU_EVENT"])
    E.update_all_fields(True)
    E.health_check(g)
    with E.EventSet(1) as es, E.PolicyRegistration(g, P["GPU_EVENT"]) as reg:
        E.diag_result_submit(1, "memtest", H["MEM"], A.HEALTH_FAIL, "memtest found 3 bad words in 257.8 GiB", value=3, ts_us=1_700_000_000_000_000)
        ev = es.wait(2000)
        assert ev and ev["code"] == A.EVENT["DIAG_FAIL"] and "memtest" in ev["message"]
        v = reg.wait(2000)
        assert v and v["data"]["code"] == A.EVENT["DIAG_FAIL"]
        # the same report read again is not a second failure
        E.diag_result_submit(1, "memtest", H["MEM"], A.HEALTH_FAIL, "memtest found 3 bad words in 257.8 GiB", value=3, ts_us=1_700_000_000_000_000)
        assert es.wait(300) is None
        # a degraded result is an incident without an event
        E.diag_result_submit(1, "xgmi_p2p", H["XGMI"], A.HEALTH_WARN, "link to GPU 4 reaches 61 % of its siblings", value=93.0)
        assert es.wait(300) is None
    r = E.health_check(g)[0]
    assert r["status"] == "Failure"
    kinds = sorted((w["type"], w["status"]) for w in r["watches"])
    assert ("Memory watches", "Failure") in kinds and ("xGMI watches", "Warning") in kinds
    assert any("Diagnostic memtest: memtest found 3 bad words" in w["error"] for w in r["watches"])
    # the verdict stays until the test passes again
    assert ("Memory watches", "Failure") in sorted((w["type"], w["status"]) for w in E.health_check(g)[0]["watches"])
    E.diag_result_submit(1, "memtest", H["MEM"], A.HEALTH_PASS, "no bad words", value=0)
    E.diag_result_submit(1, "xgmi_p2p", H["XGMI"], A.HEALTH_PASS, "all links within 5 %")
    assert E.health_check(g)[0]["status"] == "Healthy"
    assert {d["test"]: d["health"] for d in E.diag_results(1)} == {"memtest": 0, "xgmi_p2p": 0}
    for bad in (dict(system=0), dict(system=H["MEM"] | H["XGMI"]), dict(health=7), dict(gpu=99), dict(test="")):
        kw = dict(gpu=1, test="t", system=H["MEM"], health=A.HEALTH_WARN)
        kw.update(bad)
        with pytest.raises(E.GpumonError):
            E.diag_result_submit(kw["gpu"], kw["test"], kw["system"], kw["health"])


def test_exporter_hands_report_verdicts_to_the_engine(engine, tmp_path):
    rep = {"gpu": 2, "timestamp": 1_700_000_123, "result": "fail",
           "verdicts": [{"test": "gemm_bf16", "system": H["COMPUTE"], "health": A.HEALTH_FAIL, "value": 0.31,
                         "message": "GEMM bf16: sampled entries differ from the fp32 reference"},
                        {"test": "hbm_stream", "system": H["MEM"], "health": A.HEALTH_PASS, "value": 6.6, "message": "ok"}]}
    (tmp_path / "gpu2.json").write_text(json.dumps(rep))
    g = E.group_create("rep", [2])
    E.health_set(g, H["ALL"])
    E.health_check(g)
    with E.Exporter(interval_ms=1000, diag_results_path=str(tmp_path)) as ex:
        ex.sweep()
        text = ex.render()
        ex.render()
    assert 'gpumon_diag_pass{gpu="2"} 0' in text
    r = E.health_check(g)[0]
    assert ("Compute (XCD) watches", "Failure") in [(w["type"], w["status"]) for w in r["watches"]]
    assert {d["test"] for d in E.diag_results(2)} == {"gemm_bf16", "hbm_stream"}
    assert E.latest_values(2, [F["LAST_EVENT_CODE"]])[F["LAST_EVENT_CODE"]] == A.EVENT["DIAG_FAIL"]


def test_link_test_report_carries_one_verdict_per_gpu(engine, tmp_path):
    """The link test is about several GPUs at once: its report names the GPU in every verdict, and the exporter hands
    each to the engine under that GPU, not under the report's."""
    rep = {"gpu": 0, "timestamp": 1_700_000_456, "result": "fail", "p2p_gbps": [[0, 64.0], [41.0, 0]],
           "verdicts": [{"test": "xgmi_p2p", "gpu": 0, "system": H["XGMI"], "health": A.HEALTH_PASS, "value": 64.0,
                         "message": "1 links, slowest 64.0 GB/s"},
                        {"test": "xgmi_p2p", "gpu": 1, "system": H["XGMI"], "health": A.HEALTH_FAIL, "value": 41.0,
                         "message": "link 0 to GPU 0 (uni): 41.0 GB/s is 64 % of the other links' median 64.0 GB/s"}]}
    (tmp_path / "links.json").write_text(json.dumps(rep))
    with E.Exporter(interval_ms=1000, diag_results_path=str(tmp_path)) as ex:
        ex.sweep()
        ex.render()
    assert [(d["test"], d["health"]) for d in E.diag_results(0) if d["test"] == "xgmi_p2p"] == [("xgmi_p2p", A.HEALTH_PASS)]
    got = [d for d in E.diag_results(1) if d["test"] == "xgmi_p2p"]
    assert len(got) == 1 and got[0]["health"] == A.HEALTH_FAIL and "link 0 to GPU 0" in got[0]["message"]
    g = E.group_create("lnkv", [0, 1])
    E.health_set(g, H["XGMI"])
    res = {r["gpu"]: r for r in E.health_check(g)}
    assert res[1]["status"] == "Failure" and res[0]["status"] == "Healthy"


def test_diagnostic_results_are_bounded_and_their_names_plain(engine):
    """Any client of the engine may hand in verdicts.  A test name ends up in label values, JSON and fixed-size reply
    arrays: it must be a plain identifier, and a GPU keeps at most 32 of them; a known name keeps being replaced."""
    from k8s_gpu_monitor_amd._lib import GpumonError
    for bad in ('mem"test', "a b", "x/y", "tab\tname", "\u00e9"):
        with pytest.raises(GpumonError) as e:
            E.diag_result_submit(0, bad, H["MEM"], A.HEALTH_PASS)
        assert e.value.status == A.GM_ST_BADPARAM, bad
    for i in range(32):
        E.diag_result_submit(0, f"test-{i}.v1_x", H["MEM"], A.HEALTH_PASS)
    with pytest.raises(GpumonError) as e:
        E.diag_result_submit(0, "one-too-many", H["MEM"], A.HEALTH_FAIL)
    assert e.value.status == A.GM_ST_INSUFFICIENT_SIZE
    E.diag_result_submit(0, "test-7.v1_x", H["MEM"], A.HEALTH_WARN, "still replaceable")
    res = {d["test"]: d["health"] for d in E.diag_results(0)}
    assert len(res) == 32 and res["test-7.v1_x"] == A.HEALTH_WARN
    E.diag_result_submit(1, "one-too-many", H["MEM"], A.HEALTH_PASS)   # the bound is per GPU


def test_links_and_diagnostics_views_cli_and_rest(engine, tmp_path):
    """`gpumon links`, `gpumon diag`, and the REST routes /gpumon/links/id/N and /gpumon/diag/id/N (text and /json)."""
    import signal
    import socket
    import urllib.request
    text = E.render_view("links", 2)
    assert text.splitlines()[0] == "xGMI links of GPU 2" and len(text.splitlines()) == 2 + 7
    assert "Up" in text and "0000:" in text
    doc = json.loads(E.render_view("links", 2, json=True))
    assert doc["gpu"] == 2 and [l["peer_gpu"] for l in doc["links"]] == [0, 1, 3, 4, 5, 6, 7]
    assert all(l["state"] == "up" and l["max_bandwidth_gbps"] == 608 for l in doc["links"])
    E.diag_result_submit(2, "memtest", H["MEM"], A.HEALTH_FAIL, "3 words differ", value=3.0)
    d = json.loads(E.render_view("diag", 2, json=True))
    assert d["results"][0]["test"] == "memtest" and d["results"][0]["health"] == "Failure" and d["results"][0]["value"] == 3
    assert "memtest" in E.render_view("diag", 2) and "(no diagnostic has reported)" in E.render_view("diag", 5)
    with pytest.raises(E.GpumonError):
        E.render_view("links", 99)

    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=4,seed=1,steady=1")
    r = subprocess.run([bin_path("gpumon"), "links", "-g", "1", "--json"], env=env, capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and [l["peer_gpu"] for l in json.loads(r.stdout)["links"]] == [0, 2, 3]
    r = subprocess.run([bin_path("gpumon"), "diag"], env=env, capture_output=True, text=True, timeout=30)
    assert r.returncode == 0 and r.stdout.count("Diagnostics of GPU") == 4

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    sock = str(tmp_path / "he.sock")
    p = subprocess.Popen([bin_path("gpumon-hostengine"), "--domain-socket", sock, "--http", f"127.0.0.1:{port}", "--pid-file",
                          str(tmp_path / "he.pid")], env=env, stderr=subprocess.PIPE)
    try:
        for _ in range(100):
            if os.path.exists(sock):
                break
            time.sleep(0.05)
        get = lambda path: urllib.request.urlopen(f"http://127.0.0.1:{port}{path}", timeout=5).read().decode()  # noqa: E731
        assert [l["peer_gpu"] for l in json.loads(get("/gpumon/links/id/3/json"))["links"]] == [0, 1, 2]
        assert get("/gpumon/links/id/0").startswith("xGMI links of GPU 0")
        assert json.loads(get("/gpumon/diag/id/0/json")) == {"gpu": 0, "results": []}
        with pytest.raises(urllib.error.HTTPError):
            get("/gpumon/links/id/9")
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)


# ------------------------------------------------------------ link grading --
def _matrix(n, value, **cells):
    m = [[0.0 if i == j else float(value) for j in range(n)] for i in range(n)]
    for key, v in cells.items():
        i, j = (int(x) for x in key[1:].split("_"))
        m[i][j] = v
    return m


def test_link_grading_floor_and_sibling_ratio():
    """The link test's verdict arithmetic (diag/hip/link_test.hip, gmd_link_grade) on synthetic matrices: an
    8-GPU full mesh has 56 directed links; one slow lane fails against its siblings, a mildly slow one warns,
    a uniformly slow fabric fails only because of the absolute floor."""
    from k8s_gpu_monitor_amd.ops import diag
    worst, v = diag.link_grade(_matrix(8, 64.0), floor_gbps=38.0)
    assert worst == 0 and len(v) == 56 and all(x["health"] == 0 for x in v)
    assert {(x["src"], x["dst"]) for x in v} == {(i, j) for i in range(8) for j in range(8) if i != j}
    assert all(x["sibling_median_gbps"] == pytest.approx(64.0) for x in v)

    worst, v = diag.link_grade(_matrix(8, 64.0, c0_4=40.0, c3_1=56.0), floor_gbps=38.0)
    by = {(x["src"], x["dst"]): x for x in v}
    assert worst == 20
    assert by[(0, 4)]["health"] == 20 and "62 % of the other links' median 64.0" in by[(0, 4)]["message"]
    assert by[(3, 1)]["health"] == 10 and "88 %" in by[(3, 1)]["message"]
    assert sum(1 for x in v if x["health"]) == 2     # the slow lanes do not drag their siblings' median down
    assert by[(4, 0)]["health"] == 0                 # the other direction of the same link is its own verdict

    # every link equally slow: the ratio rule sees nothing, the floor does
    worst, v = diag.link_grade(_matrix(8, 30.0), floor_gbps=38.0)
    assert worst == 20 and all(x["health"] == 20 and "under the floor of 38.0" in x["message"] for x in v)
    worst, v = diag.link_grade(_matrix(8, 30.0))
    assert worst == 0

    # thresholds are configurable; fail_ratio never exceeds warn_ratio
    worst, v = diag.link_grade(_matrix(4, 60.0, c1_2=56.0), warn_ratio=0.99, fail_ratio=0.95)
    assert worst == 20 and sum(1 for x in v if x["health"] == 20) == 1


def test_link_grading_peer_access_corruption_and_small_hives():
    from k8s_gpu_monitor_amd.ops import diag
    m = _matrix(4, 60.0, c2_3=-1.0)
    bad = [[0] * 4 for _ in range(4)]
    bad[1][0] = 7
    worst, v = diag.link_grade(m, bad=bad)
    by = {(x["src"], x["dst"]): x for x in v}
    assert worst == 20
    assert by[(2, 3)]["health"] == 20 and by[(2, 3)]["message"] == "no peer access"
    assert by[(1, 0)]["health"] == 20 and by[(1, 0)]["message"] == "7 corrupt 16-byte words"
    assert sum(1 for x in v if x["health"]) == 2
    # a pair without access is not a sibling: the median is over measured links only
    assert by[(0, 1)]["sibling_median_gbps"] == pytest.approx(60.0)

    # one GPU: nothing to grade, and not an error
    assert diag.link_grade([[0.0]], floor_gbps=38.0) == (0, [])
    # two GPUs: one sibling is no population, only the floor applies
    worst, v = diag.link_grade([[0, 60.0], [20.0, 0]])
    assert worst == 0 and all(x["sibling_median_gbps"] == 0 for x in v)
    worst, v = diag.link_grade([[0, 60.0], [20.0, 0]], floor_gbps=38.0)
    assert worst == 20 and [x["health"] for x in v] == [0, 20]
    # three GPUs: the other five links are the population
    worst, v = diag.link_grade(_matrix(3, 50.0, c0_1=30.0))
    assert worst == 20 and sum(1 for x in v if x["health"]) == 1


def test_link_test_without_a_gpu_is_an_error_not_a_crash():
    import torch
    from k8s_gpu_monitor_amd.ops import diag
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(diag.DiagError):
        diag.link_test("uni", devices=[0])


def test_gemm_shape_rules_are_checked_before_anything_touches_a_device():
    """The GEMM entry points refuse a shape they cannot run before any HIP call, with a message that says what the shape
    must be: M and N in multiples of 8 for bf16 and fp8 (256 for MX-fp8, whose packed scales know no partial tile), K in
    whole 128-byte K-tiles, operands 16-byte aligned.  No GPU is needed to see that."""
    from k8s_gpu_monitor_amd.ops import diag
    p = 0x10000  # never dereferenced: the shape is refused first
    for dtype, m, n, k in [("bf16", 500, 512, 512), ("bf16", 512, 508, 512), ("bf16", 512, 512, 96), ("bf16", 0, 512, 512),
                           ("fp8", 512, 512, 192), ("fp8", 12, 512, 256)]:
        with pytest.raises(diag.DiagError) as e:
            diag.gemm_launch(dtype, p, p, p, m, n, k)
        assert "multiples of 8" in str(e.value), (dtype, m, n, k)
    with pytest.raises(diag.DiagError) as e:
        diag.gemm_launch("mxfp8", p, p, p, 264, 512, 512, scale_a=p, scale_b=p)
    assert "256 (MX-fp8)" in str(e.value)
    with pytest.raises(diag.DiagError) as e:
        diag.gemm_launch("bf16", p + 8, p, p, 512, 512, 512)
    assert "16-byte aligned" in str(e.value)
    with pytest.raises(diag.DiagError):
        diag.mx_pack_scales(p, p, 192, 512)   # rows in multiples of 128
    with pytest.raises(diag.DiagError) as e:
        diag.gemm_stress("bf16", 1000, 1001, 512, seconds=0.01)
    assert "multiples of 8" in str(e.value)


def test_gpumon_diag_takes_every_documented_value_flag():
    """Every flag of the usage text that carries a value is parsed as such (a flag missing from the parser's list
    turns its value into an 'unexpected argument').  Without a GPU the commands then stop at the device, not at
    the command line."""
    import re
    usage = subprocess.run([bin_path("gpumon-diag")], capture_output=True, text=True).stderr
    flags = set(re.findall(r"(--?[a-z][a-z-]*) [A-Z_|0-9a-z]*[A-Z]", usage))
    assert {"--submit", "--min-busbw", "--floor-gbps", "--publish", "--bytes", "-g"} <= flags, sorted(flags)
    for f in sorted(flags):
        r = subprocess.run([bin_path("gpumon-diag"), "links", f, "7"], capture_output=True, text=True, timeout=60)
        assert "unexpected argument" not in r.stderr and "missing value" not in r.stderr, (f, r.stderr)
