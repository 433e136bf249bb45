"""The real device backend (core/backend/amdsmi_backend.cc) on the CPU, against a stand-in for libamd_smi.so
(tests/stubs/amdsmi_stub.cc) and a sysfs tree that holds metrics tables captured from an MI355X.

Covers what otherwise only runs on hardware: enumeration in PCI order, absent optional entry points, the bulk
metrics table read from the driver's file after it matched the library's decoding, and the fall back to the library
when it does not match, when the layout is another one, or when the operator says so."""
import json
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = "/opt/rocm/include/amd_smi/amdsmi.h"
FIXTURE = os.path.join(ROOT, "tests", "data", "gpu_metrics_v1_8.json")

pytestmark = pytest.mark.skipif(not os.path.exists(HEADER) or not shutil.which("g++"), reason="needs the amd_smi header and g++")

PROBE = r"""
import json, sys
from k8s_gpu_monitor_amd import abi as A, engine as E
E.init(backend="amdsmi")
F = A.FI
ids = [F["GPU_TEMP"], F["POWER_USAGE"], F["GFX_CLOCK"], F["MEM_CLOCK"], F["GFX_UTIL"], F["XGMI_BANDWIDTH_TOTAL"], F["ENC_UTIL"]]
out = {"count": E.device_count(), "gpus": []}
g = E.group_create("all", list(range(E.device_count())))
fg = E.field_group_create("f", ids)
E.watch_fields(g, fg, update_freq_us=1_000_000)
for _ in range(int(sys.argv[1])):
    E.update_all_fields(True)
for i in range(E.device_count()):
    a = E.device_attributes(i)
    v = E.latest_values(i, ids)
    out["gpus"].append({"bdf": a["pci_bdf"], "uuid": a["uuid"], "name": a["name"], "num_xcd": a["num_xcd"], "table": E.metrics_table_source(i),
                        "temp": v[F["GPU_TEMP"]], "power": v[F["POWER_USAGE"]], "gfx_clock": v[F["GFX_CLOCK"]], "mem_clock": v[F["MEM_CLOCK"]],
                        "util": v[F["GFX_UTIL"]], "xgmi_kb": v[F["XGMI_BANDWIDTH_TOTAL"]], "enc": v[F["ENC_UTIL"]],
                        "procs": E.process_source(i)})
E.shutdown()
print(json.dumps(out))
"""


@pytest.fixture(scope="module")
def stub(tmp_path_factory):
    d = tmp_path_factory.mktemp("amdsmi_stub")
    lib = d / "libamd_smi_stub.so"
    subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-I/opt/rocm/include", os.path.join(ROOT, "tests", "stubs", "amdsmi_stub.cc"),
                    "-o", str(lib)], check=True, timeout=120)
    return lib


@pytest.fixture()
def tables():
    with open(FIXTURE) as f:
        return [bytes.fromhex(p["raw"]) for p in json.load(f)["pairs"]]


def run_backend(stub, tmp_path, raws, samples=3, **env):
    sysfs = tmp_path / "sys"
    for i, raw in enumerate(raws):
        dev = sysfs / "bus" / "pci" / "devices" / f"0000:{0x10 + i:02x}:00.0"
        dev.mkdir(parents=True, exist_ok=True)
        (dev / "gpu_metrics").write_bytes(raw)
        (dev / "numa_node").write_text("0\n")
    calls = tmp_path / "calls"
    e = dict(os.environ, PYTHONPATH=ROOT, GPUMON_AMDSMI_LIB=str(stub), GPUMON_SYSFS_ROOT=str(sysfs), STUB_SYSFS=str(sysfs),
             STUB_GPUS=str(len(raws)), STUB_CALLS_FILE=str(calls), GPUMON_PROC_ROOT=str(tmp_path / "noproc"), GPUMON_LOG_LEVEL="info")
    e.pop("GPUMON_BACKEND", None)
    e.update({k: str(v) for k, v in env.items()})
    r = subprocess.run([sys.executable, "-c", PROBE, str(samples)], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0, r.stderr[-800:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out["library_calls"] = calls.stat().st_size if calls.exists() else 0
    out["stderr"] = r.stderr
    return out


def test_enumeration_and_direct_table(stub, tmp_path, tables):
    idle, _, load = tables
    out = run_backend(stub, tmp_path, [idle, load], samples=5)
    assert out["count"] == 2
    a, b = out["gpus"]
    assert (a["bdf"], b["bdf"]) == ("0000:10:00.0", "0000:11:00.0") and a["uuid"] != b["uuid"]
    assert a["name"] == "AMD Instinct MI355X" and a["num_xcd"] == 8
    assert a["table"] == b["table"] == "direct"
    # GPU 0 holds the idle table, GPU 1 the one taken under a GEMM soak
    assert a["power"] < 400 and a["util"] == 0 and a["gfx_clock"] < 200
    assert b["power"] >= 1300 and b["util"] == 100 and 1500 < b["gfx_clock"] < 2400 and b["mem_clock"] == 2000
    assert 40 <= a["temp"] <= b["temp"] <= 110
    assert a["enc"] is None  # no encoder on this part: blank, not zero
    assert a["xgmi_kb"] > 0
    # the library was asked for the table while enumerating and checking, never for a sample:
    # two reads per GPU at start (XCD count, comparison), none for the five sample cycles
    assert out["library_calls"] <= 2 * 2 + 2, out["library_calls"]
    assert a["procs"] == "none"  # no KFD table, no /proc tree: says so instead of an empty list that looks like "idle"


def test_a_library_that_decodes_differently_wins(stub, tmp_path, tables):
    out = run_backend(stub, tmp_path, [tables[2]], samples=4, STUB_SKEW=3)
    g = out["gpus"][0]
    assert g["table"].startswith("library: differs from the library: temperature_hotspot"), g["table"]
    # the values are the library's (skewed by 3), and every sample went through it
    direct = run_backend(stub, tmp_path, [tables[2]], samples=4)["gpus"][0]
    assert direct["table"] == "direct" and g["temp"] == direct["temp"] + 3
    assert out["library_calls"] >= 4 + 2


def test_other_layout_or_operator_choice_keeps_the_library(stub, tmp_path, tables):
    out = run_backend(stub, tmp_path, [tables[0]], GPUMON_METRICS_TABLE="library")
    assert out["gpus"][0]["table"] == "library: switched off by GPUMON_METRICS_TABLE" and out["gpus"][0]["temp"] > 0
    # a table of another revision: the stub library refuses it as well, so values are blank, and the reason is kept
    v17 = bytearray(tables[0])
    v17[3] = 7
    out = run_backend(stub, tmp_path, [bytes(v17)])
    assert out["gpus"][0]["table"] == "library: table format 1.7 (this decoder reads 1.8)", out["gpus"][0]["table"]
    assert out["gpus"][0]["temp"] is None


PROBE_OPTIONAL = r"""
import json, os, sys, time
from k8s_gpu_monitor_amd import abi as A, engine as E
script = os.environ["STUB_SCRIPT"]
def device(**kv):
    with open(script + ".tmp", "w") as f:
        f.write("".join(f"{k}={v}\n" for k, v in kv.items()))
    os.replace(script + ".tmp", script)
device(ecc_ce=5, ecc_ue=0, vram_used_mib=2048, bad_reserved=2, bad_pending=1, power_cap_w=1200)
E.init(backend="amdsmi")
F = A.FI
ids = [F["ECC_CE_VOLATILE"], F["ECC_UE_VOLATILE"], F["ECC_UMC_UE"], F["VRAM_USED"], F["VRAM_FREE"], F["VRAM_TOTAL"], F["BAD_PAGES_UE"],
       F["BAD_PAGES_PENDING"], F["BAD_PAGES_CE"], F["XGMI_ERROR_STATUS"], F["LAST_EVENT_CODE"]]
g = E.group_create("g", [0]); fg = E.field_group_create("f", ids)
E.watch_fields(g, fg, update_freq_us=1_000_000)
out = {}
def snap(name):
    E.update_all_fields(True)
    raw = E.latest_values(0, ids, raw=True)
    out[name] = {k: (raw[i].py(), raw[i].status) for i, k in enumerate(["ce", "ue", "umc_ue", "vram_used", "vram_free", "vram_total", "bad_ue",
                                                                           "bad_pending", "bad_ce", "xgmi_err", "event"])}
snap("first")
E.health_set(g)
E.health_check(g)  # the baseline the next check compares with
device(ecc_ce=5, ecc_ue=2, vram_used_mib=4096, bad_reserved=3, bad_pending=0, power_cap_w=1200)
snap("after_ue")
time.sleep(0.2)  # the event reaches the field through the notifier thread
snap("after_event")
h = E.health_check(g)[0]
out["health"] = {"status": h["status"], "overall": h["overall"], "incidents": h.get("incidents")}
device(ecc_status=10)
snap("no_perm")
device(ecc_ce=5, ecc_ue=2, vram_used_mib=0)
snap("vram_zero")
E.shutdown()
print(json.dumps(out, default=str))
"""


def test_optional_sources_and_the_ras_chain_through_the_real_backend(stub, tmp_path, tables):
    sysfs = tmp_path / "sys"
    dev = sysfs / "bus" / "pci" / "devices" / "0000:10:00.0"
    dev.mkdir(parents=True)
    (dev / "gpu_metrics").write_bytes(tables[0])
    (dev / "mem_info_vram_used").write_text("0\n")
    from k8s_gpu_monitor_amd import abi as A
    e = dict(os.environ, PYTHONPATH=ROOT, GPUMON_AMDSMI_LIB=str(stub), GPUMON_SYSFS_ROOT=str(sysfs), STUB_SYSFS=str(sysfs), STUB_GPUS="1",
             STUB_SCRIPT=str(tmp_path / "script"), GPUMON_PROC_ROOT=str(tmp_path / "noproc"))
    e.pop("GPUMON_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", PROBE_OPTIONAL], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0, r.stderr[-800:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    first, after = out["first"], out["after_event"]
    assert first["ce"][0] == 5 and first["ue"][0] == 0 and first["umc_ue"][0] == 0
    assert first["vram_used"][0] == 2048 and first["vram_total"][0] == 294896 and first["vram_free"][0] == 294896 - 2048
    assert first["bad_ue"][0] == 2 and first["bad_pending"][0] == 1
    assert first["bad_ce"] == [None, A.GM_VS_NOT_SUPPORTED]      # the driver does not say why a page was retired
    assert first["xgmi_err"] == [None, A.GM_VS_NOT_SUPPORTED]    # entry point absent from the library
    assert first["event"][0] in (None, 0)
    # two uncorrectable errors between two samples: the counters move, and the step becomes a RAS event
    assert out["after_ue"]["ue"][0] == 2 and out["after_ue"]["umc_ue"][0] == 2 and out["after_ue"]["bad_ue"][0] == 3
    assert after["event"][0] == A.EVENT["RAS_UE"], after
    assert out["health"]["overall"] == 20, out["health"]  # uncorrectable memory errors and a growing bad-page table: failure
    # the library refusing the RAS calls blanks those fields with the reason, and nothing else
    np_ = out["no_perm"]
    assert np_["ue"] == [None, A.GM_VS_NO_PERMISSION] and np_["umc_ue"] == [None, A.GM_VS_NO_PERMISSION]
    assert np_["vram_total"][0] == 294896
    # 0 bytes used with the kernel counter present is a reading (the file exists in this tree)
    assert out["vram_zero"]["vram_used"][0] == 0


PROBE_PARTITIONS = r"""
import json
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd.utils.promparse import parse
E.init(backend="amdsmi")
out = {"devices": [(E.device_attributes(i)["pci_bdf"], E.device_attributes(i)["uuid"], E.device_mode(i)["compute_partition"]) for i in range(E.device_count())]}
with E.Exporter(interval_ms=1000) as ex:
    ex.sweep()
    fams = parse(ex.render())
out["labels"] = [l for l, _ in fams["dcgm_sm_clock"].samples]
E.shutdown()
print(json.dumps(out))
"""


def test_partitions_are_enumerated_in_order_and_labelled(stub, tmp_path, tables):
    """A GPU in CPX-like mode shows up once per partition under one PCI address.  The library hands the partitions out
    in whatever order; the backend sorts by address, then partition, and the exporter says which slice a series is."""
    sysfs = tmp_path / "sys"
    for i in range(2):
        dev = sysfs / "bus" / "pci" / "devices" / f"0000:{0x10 + i:02x}:00.0"
        dev.mkdir(parents=True)
        (dev / "gpu_metrics").write_bytes(tables[i])
    e = dict(os.environ, PYTHONPATH=ROOT, GPUMON_AMDSMI_LIB=str(stub), GPUMON_SYSFS_ROOT=str(sysfs), STUB_SYSFS=str(sysfs), STUB_GPUS="2",
             STUB_PARTITIONS="4", GPUMON_PROC_ROOT=str(tmp_path / "noproc"))
    e.pop("GPUMON_BACKEND", None)
    r = subprocess.run([sys.executable, "-c", PROBE_PARTITIONS], capture_output=True, text=True, timeout=120, env=e)
    assert r.returncode == 0, r.stderr[-800:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert [d[0] for d in out["devices"]] == ["0000:10:00.0"] * 4 + ["0000:11:00.0"] * 4
    assert len({d[1] for d in out["devices"]}) == 8 and all(d[2] == "QPX" for d in out["devices"])
    # index order = address, then partition 0..3, although the library listed 3..0
    assert [d[1][-8:-4] for d in out["devices"]] == ["0000", "0001", "0002", "0003"] * 2
    assert [l["gpu"] for l in out["labels"]] == [str(i) for i in range(8)]
    assert [l["partition"] for l in out["labels"]] == ["0", "1", "2", "3"] * 2


def test_shutdown_waits_for_the_calls_the_backend_has_inside_the_library(stub, tmp_path, tables):
    """The first read of the throttle state starts a one-time cross-check on a thread of its own, because the library's
    violation report sleeps 100 ms inside.  A short-lived caller shuts down well within that: the backend must wait for
    its thread before it closes the library (on an MI355X host a caller that did not was left with a damaged heap one
    time in four while another process kept the library busy).  The stand-in library stays inside for 300 ms and ends
    the process if it is shut down meanwhile."""
    idle = tables[0]
    sysfs = tmp_path / "sys"
    dev = sysfs / "bus" / "pci" / "devices" / "0000:10:00.0"
    dev.mkdir(parents=True)
    (dev / "gpu_metrics").write_bytes(idle)
    (dev / "numa_node").write_text("0\n")
    e = dict(os.environ, PYTHONPATH=ROOT, GPUMON_AMDSMI_LIB=str(stub), GPUMON_SYSFS_ROOT=str(sysfs), STUB_SYSFS=str(sysfs), STUB_GPUS="1",
             GPUMON_PROC_ROOT=str(tmp_path / "noproc"), STUB_VIOLATION_MS="300")
    e.pop("GPUMON_BACKEND", None)
    code = ("import time; from k8s_gpu_monitor_amd import smi; t0 = time.time(); smi.init(backend='amdsmi'); s = smi.device_status(0); "
            "t1 = time.time(); smi.shutdown(); print(s['power_w'], round(t1 - t0, 3), round(time.time() - t1, 3))")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode == 0, (r.returncode, r.stderr[-600:])
    power, read_s, shutdown_s = r.stdout.split()
    assert float(power) > 0 and float(read_s) < 0.25, r.stdout    # the read did not wait for the check ...
    assert 0.05 < float(shutdown_s) < 2.0, r.stdout               # ... the shutdown did
