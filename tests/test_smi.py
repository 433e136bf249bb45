"""Direct device layer (gm_smi_*) and the nvml-shaped data model on the fake
backend, plus the amd-smi oracle's parser on canned output.

Reference API being mirrored: bindings/go/nvml/nvml.go:274-604 and its tests
nvml_test.go:18-218 (device count, driver version, device info, device status)."""
import threading
import time

import pytest

from k8s_gpu_monitor_amd import abi as A
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd import smi
from k8s_gpu_monitor_amd.models import device as M
from k8s_gpu_monitor_amd.utils import amdsmi_cli

F = A.FI


@pytest.fixture
def direct():
    smi.init(backend="fake", backend_spec="gpus=4,seed=5,steady=1")
    yield
    smi.shutdown()


def test_lifetime_is_refcounted_and_independent_of_the_engine():
    with pytest.raises(E.GpumonError) as e:
        smi.device_count()
    assert e.value.status == A.GM_ST_UNINITIALIZED
    with pytest.raises(E.GpumonError):
        smi.shutdown()
    smi.init(backend="fake", backend_spec="gpus=2")
    smi.init(backend="fake", backend_spec="gpus=7")  # second init only takes a reference
    assert smi.device_count() == 2
    with pytest.raises(E.GpumonError):
        E.device_count()  # the engine was never started
    smi.shutdown()
    assert smi.device_count() == 2
    smi.shutdown()
    with pytest.raises(E.GpumonError):
        smi.device_count()
    with pytest.raises(E.GpumonError) as e:
        smi.init(backend="fake", backend_spec="fail_init=1")
    assert e.value.status == A.GM_ST_INIT_ERROR


def test_direct_reads(direct):
    assert smi.device_count() == 4
    assert smi.driver_version().startswith("6.")
    assert smi.runtime_version() == (7, 2, "7.2.0")  # the CUDA-driver-version analog
    a = smi.device_attributes(2)
    assert a["name"] == "AMD Instinct MI355X" and a["uuid"].endswith("02") and a["num_cu"] == 256
    s = smi.device_status(2)
    assert 30 <= s["temperature_c"] <= 110 and s["power_w"] > 50 and s["enc_util"] is None
    v = smi.read_fields(2, [F["GPU_TEMP"], F["DEV_NAME"], F["VRAM_TOTAL"], F["ENC_UTIL"], 9999])
    assert v[F["GPU_TEMP"]] == s["temperature_c"]  # steady backend: same value on every read
    assert v[F["DEV_NAME"]] == "AMD Instinct MI355X" and v[F["VRAM_TOTAL"]] == 294896
    assert v[F["ENC_UTIL"]] is None and v[9999] is None
    # no cache: a scripted change is visible on the very next read, with no engine tick in between
    smi.fake_ctl("set gpu=2 field=150 value=91")
    assert smi.device_status(2)["temperature_c"] == 91
    assert smi.device_status(1)["temperature_c"] != 91
    with pytest.raises(E.GpumonError):
        smi.device_status(4)
    m = smi.device_mode(0)
    assert m["compute_partition"] == "SPX" and m["accounting_enabled"] == 0


def test_unsupported_and_failing_devices(direct):
    smi.fake_ctl("unsupported gpu=1 field=155")
    assert smi.device_status(1)["power_w"] is None  # NOT_SUPPORTED -> None, like the reference's nil pointers
    smi.fake_ctl("fail gpu=3")
    s = smi.device_status(3)
    assert s["temperature_c"] is None and s["gfx_clock_mhz"] is None  # a failed round blanks, never raises
    smi.fake_ctl("fail gpu=3 sources=0")
    assert smi.device_status(3)["temperature_c"] is not None


def test_processes_topology_events(direct):
    assert smi.device_processes(0) == []
    smi.fake_ctl(f"proc_add gpu=0 pid=4242 name=trainer vram={2048 << 20}")
    p = smi.device_processes(0)
    assert len(p) == 1 and p[0]["pid"] == 4242 and p[0]["name"] == "trainer" and p[0]["vram_bytes"] == 2048 << 20
    link = smi.peer_link(0, 3)
    assert link["link_name"] == "XGMI1" and link["p2p_accessible"] == 1 and link["pci_bdf"] == smi.device_attributes(3)["pci_bdf"]
    with pytest.raises(E.GpumonError):
        smi.peer_link(1, 1)
    assert len(smi.device_topology(1)["peers"]) == 3
    smi.fake_ctl("topology_unsupported")
    assert smi.device_topology(1)["peers"] == []
    # events: timeout, then delivery to a blocked waiter
    t0 = time.time()
    assert smi.event_wait(150) is None and time.time() - t0 < 2
    got = []
    th = threading.Thread(target=lambda: got.append(smi.event_wait(5000)))
    th.start()
    time.sleep(0.1)
    smi.fake_ctl("event gpu=2 code=5 msg=ras-poison")
    th.join(6)
    assert got and got[0]["gpu"] == 2 and got[0]["code"] == 5 and got[0]["uuid"].endswith("02")


def test_data_model_on_both_layers(direct):
    """The same nvml-shaped Device / DeviceStatus from the direct layer and from the engine cache."""
    d = M.new_device(1, api=smi)
    assert d.model == "AMD Instinct MI355X" and d.path.startswith("/dev/dri/renderD") and d.memory == 294896
    assert d.power == 1400 and d.clocks.cores == 2400 and d.pci.bandwidth == 63008  # Gen5 x16: 3938 MB/s per lane
    assert [t.link for t in d.topology] == ["XGMI1"] * 3
    assert d.gfx_target == "gfx950" and d.gfx_version == (9, 5, 0)  # the compute-capability analog
    import dataclasses
    assert dataclasses.replace(d, gfx_target="gfx90a:sramecc+:xnack-").gfx_version == (9, 0, 10)
    assert dataclasses.replace(d, gfx_target=None).gfx_version is None
    assert d.processes() == []
    lite = M.new_device(1, lite=True, api=smi)
    assert lite.uuid == d.uuid and lite.power is None and lite.memory is None and lite.topology == []
    st = d.status()
    assert st.utilization.encoder is None and st.temperature and st.throttle_reasons() == ["No clocks throttling"]
    E.init(backend="fake", backend_spec="gpus=4,seed=5,steady=1")
    try:
        de = M.new_device(1)
        assert (de.uuid, de.model, de.memory, de.pci.bus_id) == (d.uuid, d.model, d.memory, d.pci.bus_id)
        assert de.status().temperature == st.temperature  # same seed, steady telemetry
    finally:
        E.shutdown()


def test_direct_reads_are_thread_safe(direct):
    errs = []

    def worker(g):
        try:
            for _ in range(300):
                s = smi.device_status(g)
                assert s["temperature_c"] is not None
                smi.read_fields(g, [F["POWER_USAGE"], F["GFX_CLOCK"]])
        except Exception as e:  # noqa: BLE001
            errs.append(e)

    ts = [threading.Thread(target=worker, args=(g % 4,)) for g in range(8)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs


AMDSMI_METRIC = """
WARNING: something on stderr-ish first line
{
    "gpu_data": [
        {"gpu": 0,
         "usage": {"gfx_activity": {"value": 7, "unit": "%"}, "umc_activity": {"value": 0, "unit": "%"}, "mm_activity": "N/A"},
         "power": {"socket_power": {"value": 231, "unit": "W"}, "gfx_voltage": "N/A"},
         "clock": {"gfx_0": {"clk": {"value": 132, "unit": "MHz"}, "clk_locked": "DISABLED"}},
         "temperature": {"hotspot": {"value": 41, "unit": "C"}, "mem": {"value": "38", "unit": "C"}},
         "mem_usage": {"total_vram": {"value": 294896, "unit": "MB"}, "used_vram": {"value": 0, "unit": "MB"}}}
    ]
}
"""


def test_oracle_parser_on_canned_output():
    g = [amdsmi_cli.flatten(x) for x in amdsmi_cli.parse(AMDSMI_METRIC)]
    assert len(g) == 1
    d = g[0]
    assert d["usage.gfx_activity"] == 7 and d["usage.mm_activity"] is None
    assert d["power.socket_power"] == 231 and d["temperature.mem"] == 38
    assert d["clock.gfx_0.clk"] == 132 and d["clock.gfx_0.clk_locked"] == "DISABLED"
    assert amdsmi_cli.first(d, "socket_power", "average_socket_power") == 231
    assert amdsmi_cli.first(d, "nonexistent") is None
    assert amdsmi_cli.parse('[{"gpu": 0, "bdf": "0000:05:00.0"}]')[0]["bdf"] == "0000:05:00.0"
    with pytest.raises(ValueError):
        amdsmi_cli.parse("no json here")
