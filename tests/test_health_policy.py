"""Health checks, policies, events and per-process accounting on the fake backend."""
import threading
import time

import pytest

from k8s_gpu_monitor_amd import abi as A
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd._lib import lib

F = A.FI
H = A.HEALTH
P = A.POLICY


def _health(group):
    E.update_all_fields(True)
    return {r["gpu"]: r for r in E.health_check(group)}


def _types(resp):
    return sorted((w["type"], w["status"]) for w in resp["watches"])


# --------------------------------------------------------------------- health --
def test_health_requires_set(engine):
    g = E.group_create("h", [0])
    with pytest.raises(E.GpumonError) as e:
        E.health_check(g)
    assert e.value.status == A.GM_ST_NOT_CONFIGURED


def test_healthy_node(engine):
    E.health_set(A.GM_GROUP_ALL_GPUS)
    r = _health(A.GM_GROUP_ALL_GPUS)
    assert sorted(r) == list(range(8))
    assert all(x["status"] == "Healthy" and not x["watches"] for x in r.values())


def test_health_incidents_by_subsystem(engine):
    g = E.group_create("h", [0, 1, 2, 3])
    E.health_set(g, H["ALL"])
    assert all(x["status"] == "Healthy" for x in _health(g).values())
    E.fake_ctl("ecc gpu=0 block=0 ue=2")        # HBM uncorrectable -> memory failure
    E.fake_ctl("link_down gpu=1 link=3")        # xGMI link down -> failure
    E.fake_ctl("pcie_replay gpu=2 count=5")     # PCIe replays -> warning
    E.fake_ctl("set gpu=3 field=150 value=104")  # hot -> thermal warning
    r = _health(g)
    # the step of the uncorrectable total is also a RAS event (the XID analog): memory AND driver watch fail
    assert r[0]["status"] == "Failure" and _types(r[0]) == [("Driver / RAS event watches", "Failure"), ("Memory watches", "Failure")]
    assert "2 uncorrectable HBM" in " ".join(w["error"] for w in r[0]["watches"])
    assert "RAS uncorrectable error" in " ".join(w["error"] for w in r[0]["watches"])
    assert r[1]["status"] == "Failure" and ("xGMI watches", "Failure") in _types(r[1])
    assert "link 3 is down" in " ".join(w["error"] for w in r[1]["watches"])
    assert r[2]["status"] == "Warning" and _types(r[2]) == [("PCIe watches", "Warning")]
    assert r[3]["status"] == "Warning" and _types(r[3]) == [("Thermal watches", "Warning")]
    # counters report what happened SINCE the last check; levels keep reporting
    r = _health(g)
    assert r[0]["status"] == "Healthy" and r[2]["status"] == "Healthy"
    assert r[1]["status"] == "Failure" and r[3]["status"] == "Warning"
    E.fake_ctl("link_up gpu=1 link=3")
    E.fake_ctl("set gpu=3 field=150 value=111")
    r = _health(g)
    assert r[1]["status"] == "Healthy"
    assert r[3]["status"] == "Failure"  # shutdown region


def test_health_more_subsystems(engine):
    g = E.group_create("h2", [0, 1, 2, 3, 4])
    E.health_set(g, H["ALL"])
    _health(g)
    E.fake_ctl("ecc gpu=0 block=1 ue=1")            # GFX block -> compute
    E.fake_ctl("bad_pages gpu=1 pending=3")
    E.fake_ctl("violation gpu=2 power=5000 thermal=700")
    E.fake_ctl("event gpu=3 code=3 msg=reset")      # GPU reset -> driver failure
    E.fake_ctl("ecc gpu=4 block=4 ce=7")            # xGMI correctable -> warning
    time.sleep(0.3)  # the event travels through the event thread
    r = _health(g)
    assert _types(r[0]) == [("Compute (XCD) watches", "Failure"), ("Driver / RAS event watches", "Failure")]
    assert _types(r[1]) == [("Memory watches", "Warning")] and "pending retirement" in r[1]["watches"][0]["error"]
    assert _types(r[2]) == [("Power watches", "Warning"), ("Thermal watches", "Warning")]
    assert _types(r[3]) == [("Driver / RAS event watches", "Failure")]
    assert _types(r[4]) == [("xGMI watches", "Warning")]


def test_health_mask_limits_subsystems(engine):
    g = E.group_create("h3", [0])
    E.health_set(g, H["THERMAL"])
    _health(g)
    E.fake_ctl("ecc gpu=0 block=0 ue=1")
    E.fake_ctl("set gpu=0 field=150 value=105")
    r = _health(g)[0]
    assert _types(r) == [("Thermal watches", "Warning")]
    v = __import__("ctypes").c_uint32()
    assert lib().gm_health_get(g, v) == 0 and v.value == H["THERMAL"]


def test_health_view_matches_rest_shape(engine):
    import json
    j = json.loads(E.render_view("health", 2, json=True))
    assert j == {"GPU": 2, "Status": "Healthy", "Watches": []}
    assert "Status             : Healthy" in E.render_view("health", 2)


# --------------------------------------------------------------------- policy --
def test_policy_defaults_and_set(engine):
    p = E.policy_get(A.GM_GROUP_ALL_GPUS)
    assert (p["max_bad_pages"], p["thermal_c"], p["power_w"]) == (10, 100, 1400)
    E.policy_set(A.GM_GROUP_ALL_GPUS, P["THERMAL"], thermal_c=80)
    p = E.policy_get(A.GM_GROUP_ALL_GPUS)
    assert p["thermal_c"] == 80 and p["power_w"] == 1400  # only the named condition changed


def test_thermal_policy_fires_once_per_crossing_with_hysteresis(engine):
    g = E.group_create("p", [0, 1])
    E.policy_set(g, P["THERMAL"], thermal_c=90, hysteresis_pct=10)
    with E.PolicyRegistration(g, P["THERMAL"]) as reg:
        assert reg.wait(200) is None
        E.fake_ctl("set gpu=1 field=150 value=95")
        E.update_all_fields(True)
        v = reg.wait(2000)
        assert v["name"] == "Thermal Limit" and v["gpu"] == 1 and v["data"] == {"temperature_c": 95}
        assert abs(v["ts_us"] / 1e6 - time.time()) < 5
        for _ in range(3):  # still hot: no repeat
            E.update_all_fields(True)
        assert reg.wait(200) is None
        E.fake_ctl("set gpu=1 field=150 value=85")  # below the limit but inside the hysteresis band
        E.update_all_fields(True)
        E.fake_ctl("set gpu=1 field=150 value=93")
        E.update_all_fields(True)
        assert reg.wait(200) is None
        E.fake_ctl("set gpu=1 field=150 value=70")  # re-arms
        E.update_all_fields(True)
        E.fake_ctl("set gpu=1 field=150 value=91")
        E.update_all_fields(True)
        assert reg.wait(2000)["data"] == {"temperature_c": 91}


def test_every_condition_delivers_its_payload(engine):
    g = E.group_create("all", [3])
    E.policy_set(g, P["ALL"], max_bad_pages=4, thermal_c=95, power_w=900)
    got = {}
    with E.PolicyRegistration(g, P["ALL"]) as reg:
        E.fake_ctl("ecc gpu=3 block=0 ue=1")
        E.fake_ctl("pcie_replay gpu=3 count=2")
        E.fake_ctl("bad_pages gpu=3 ue=5")
        E.fake_ctl("set gpu=3 field=150 value=99")
        E.fake_ctl("load gpu=3 util=90")           # ~1220 W on the fake model
        E.fake_ctl("link_down gpu=3 link=2")
        E.fake_ctl("event gpu=3 code=1 msg=vmfault")
        E.update_all_fields(True)
        deadline = time.time() + 5
        events = []
        while (len(got) < 7 or len(events) < 2) and time.time() < deadline:
            v = reg.wait(500)
            if v:
                got[v["condition"]] = v
                if v["condition"] == P["GPU_EVENT"]:
                    events.append(v["data"])
    assert set(got) == {P["ECC_UE"], P["PCIE"], P["MAX_BAD_PAGES"], P["THERMAL"], P["POWER"], P["XGMI"], P["GPU_EVENT"]}
    assert got[P["ECC_UE"]]["data"] == {"location": "HBM (UMC)", "count": 1}
    assert got[P["PCIE"]]["data"] == {"replay_counter": 2}
    assert got[P["MAX_BAD_PAGES"]]["data"] == {"ce_pages": 0, "ue_pages": 5}
    assert got[P["POWER"]]["data"]["power_w"] >= 900
    assert got[P["XGMI"]]["data"]["link"] == 2
    # two GPU events: the injected VM fault, and the RAS event the engine derives from the uncorrectable-error step
    assert sorted(e["code"] for e in events) == [1, 5]
    assert {"code": 1, "message": "vmfault"} in events
    assert [e["message"] for e in events if e["code"] == 5] == ["RAS: 1 uncorrectable error(s) since the last sample (total 1)"]


def test_policy_callback_thread_and_unregister(engine):
    g = E.group_create("cb", [0])
    seen = []
    tid = []

    def cb(v):
        seen.append(v)
        tid.append(threading.get_ident())

    reg = E.PolicyRegistration(g, P["PCIE"], callback=cb)
    E.fake_ctl("pcie_replay gpu=0 count=1")
    E.update_all_fields(True)
    deadline = time.time() + 3
    while not seen and time.time() < deadline:
        time.sleep(0.01)
    assert len(seen) == 1 and tid[0] != threading.get_ident()  # notifier thread, not ours
    reg.close()
    E.fake_ctl("pcie_replay gpu=0 count=1")
    E.update_all_fields(True)
    time.sleep(0.2)
    assert len(seen) == 1  # nothing after unregister
    assert lib().gm_policy_unregister(12345) == A.GM_ST_NOT_FOUND


def test_policy_survives_group_destroy_and_preexisting_errors(engine):
    """The reference destroys the group while the policy is live and loses it
    (bindings/go/dcgm/policy.go:387); here the registration stays armed.  Errors
    that predate the registration are not violations of it."""
    E.fake_ctl("ecc gpu=2 block=0 ue=9")
    g = E.group_create("tmp", [2])
    reg = E.PolicyRegistration(g, P["ECC_UE"])
    E.group_destroy(g)
    E.update_all_fields(True)
    assert reg.wait(200) is None
    E.fake_ctl("ecc gpu=2 block=0 ue=1")
    E.update_all_fields(True)
    assert reg.wait(2000)["data"]["count"] == 1
    reg.close()


def test_injected_value_trips_policy(engine):
    g = E.group_create("inj", [0])
    E.policy_set(g, P["POWER"], power_w=1000)
    with E.PolicyRegistration(g, P["POWER"]) as reg:
        E.inject_value(0, F["POWER_USAGE"], 1234.0, hold_ms=200)
        v = reg.wait(2000)
        assert v and v["data"] == {"power_w": 1234}


# --------------------------------------------------------------------- events --
def test_event_sets(engine):
    es = E.EventSet()  # all GPUs, critical events
    only2 = E.EventSet(gpu=2, mask=1 << A.EVENT["GPU_PRE_RESET"])
    assert es.wait(100) is None
    E.fake_ctl("event gpu=5 code=1 msg=page-fault")
    E.fake_ctl("event gpu=2 code=3")
    E.fake_ctl("event gpu=2 code=20")  # process start: not critical
    a, b = es.wait(2000), es.wait(2000)
    assert {(a["gpu"], a["code"]), (b["gpu"], b["code"])} == {(5, 1), (2, 3)}
    assert a["uuid"].startswith("GPU-") and es.wait(150) is None
    e = only2.wait(2000)
    assert (e["gpu"], e["code"]) == (2, 3) and only2.wait(100) is None
    # the XID-analog field latches the last critical code
    assert E.latest_values(2, [F["LAST_EVENT_CODE"]])[F["LAST_EVENT_CODE"]] == 3
    assert E.latest_values(0, [F["LAST_EVENT_CODE"]])[F["LAST_EVENT_CODE"]] == 0
    es.close()
    only2.close()


# ----------------------------------------------------------------- accounting --
def test_pid_accounting(engine):
    with pytest.raises(E.GpumonError) as e:
        E.pid_info(777)
    assert e.value.status == A.GM_ST_NOT_WATCHED
    E.fake_ctl("load gpu=1 util=60")
    E.watch_pid_fields(update_freq_us=50_000)
    with pytest.raises(E.GpumonError) as e:
        E.pid_info(777)
    assert e.value.status == A.GM_ST_NO_DATA
    t0 = time.time()
    E.fake_ctl("proc_add gpu=1 pid=777 name=trainer vram=2147483648")
    E.fake_ctl("proc_add gpu=1 pid=778 name=other vram=1073741824")
    E.fake_ctl("proc_add gpu=4 pid=777 name=trainer vram=4294967296")
    time.sleep(0.6)
    E.fake_ctl("ecc gpu=1 block=0 ce=3")
    E.fake_ctl("violation gpu=1 power=2500")
    E.fake_ctl("event gpu=1 code=1")
    time.sleep(0.4)
    info = {s["gpu"]: s for s in E.pid_info(777)}
    assert sorted(info) == [1, 4]
    s = info[1]
    assert s["name"] == "trainer" and s["end_ts_us"] == 0 and abs(s["start_ts_us"] / 1e6 - t0) < 1
    assert s["max_vram_bytes"] == 2 << 30
    assert 55 <= s["gfx_util_avg"] <= 65
    assert 25 <= s["proc_gfx_util_avg"] <= 35  # two processes share the engine evenly on the fake
    assert s["ecc_ce"] == 3 and s["ecc_ue"] == 0 and s["power_violation_us"] == 2500
    assert s["num_critical_events"] == 1
    # ~860 W for ~1 s, half of it attributed to this process
    assert 100_000 < s["energy_mj"] < 900_000
    assert info[4]["max_vram_bytes"] == 4 << 30 and info[4]["energy_mj"] < s["energy_mj"]
    E.fake_ctl("proc_del gpu=1 pid=777")
    time.sleep(0.3)
    s = {x["gpu"]: x for x in E.pid_info(777)}[1]
    assert s["end_ts_us"] > s["start_ts_us"]
    text = E.render_view("process", 777)
    assert "Max GPU Memory Used (bytes)  : 2147483648" in text and "Name                         : trainer" in text
    assert E.device_mode(1)["accounting_enabled"] == 1
