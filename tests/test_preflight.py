"""`gpumon-diag preflight`: the deployment checks (nothing is loaded on the GPUs; the node is only looked at).  On the
fake backend, with faults seeded through the backend's `at_start` commands: the verdicts name GPU, concern and health
subsystem, and the report is a node report the exporter can publish."""
import json
import os
import subprocess

import pytest

from conftest import bin_path

DIAG = bin_path("gpumon-diag")
pytestmark = pytest.mark.skipif(not os.path.exists(DIAG), reason="needs the built gpumon-diag")
PASS, WARN, FAIL = 0, 10, 20
PCIE, XGMI, MEM, POWER, DRIVER, FIRMWARE = 1, 2, 4, 32, 64, 128


def preflight(spec="gpus=8", *args, env=None):
    e = dict(os.environ, GPUMON_BACKEND_SPEC=spec, **(env or {}))
    e.pop("GPUMON_BACKEND", None)
    r = subprocess.run([DIAG, "preflight", "--backend", "fake", "--no-runtime", "--json", *args], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode in (0, 1), r.stderr[-400:]
    return r.returncode, json.loads(r.stdout), r.stderr


def by_gpu(doc, test):
    return {v.get("gpu"): v for v in doc["verdicts"] if v["test"] == test}


def test_a_clean_node_passes_every_check_on_every_gpu():
    rc, doc, _ = preflight()
    assert rc == 0 and doc["result"] == "pass" and doc["scope"] == "node" and doc["node"]["gpus"] == 8
    assert len(doc["devices"]) == 8 and doc["devices"][3]["pci_bdf"] == "0000:35:00.0" and doc["devices"][0]["power_cap_w"] == 1400
    for test, system in (("preflight_access", DRIVER), ("preflight_memory", MEM), ("preflight_power", POWER), ("preflight_uniform", FIRMWARE),
                         ("preflight_fabric", XGMI)):
        v = by_gpu(doc, test)
        assert sorted(v) == list(range(8)), (test, sorted(v))
        assert all(x["health"] == PASS and x["system"] == system and x["pci_bdf"] == doc["devices"][g]["pci_bdf"] for g, x in v.items())
    assert all(d["xgmi_links_up"] == 7 and d["xgmi_links_wired"] == 7 for d in doc["devices"])
    node = by_gpu(doc, "preflight_runtime")
    assert list(node) == [None] and node[None]["health"] == PASS and node[None]["system"] == DRIVER


def test_seeded_faults_are_named_with_gpu_concern_and_subsystem():
    spec = ("gpus=8,at_start=bad_pages+gpu=2+pending=3;static+gpu=5+vbios=113-OLD+power_cap=1000;ecc+gpu=6+ue=2;"
            "static+gpu=1+compute_partition=CPX;link_down+gpu=3+link=4")
    rc, doc, err = preflight(spec)
    assert rc == 1 and doc["result"] == "fail"
    mem, power, uniform = by_gpu(doc, "preflight_memory"), by_gpu(doc, "preflight_power"), by_gpu(doc, "preflight_uniform")
    assert mem[2]["health"] == WARN and "3 page(s) wait for retirement" in mem[2]["message"]
    assert mem[6]["health"] == FAIL and "2 uncorrectable" in mem[6]["message"] and mem[6]["pci_bdf"] == "0000:65:00.0"
    assert power[5]["health"] == WARN and "1000 W is under the default 1400 W" in power[5]["message"]
    assert uniform[5]["health"] == WARN and "113-OLD" in uniform[5]["message"] and "113-M3550101-100" in uniform[5]["message"]
    assert uniform[1]["health"] == WARN and "compute partition CPX" in uniform[1]["message"]
    fabric = by_gpu(doc, "preflight_fabric")
    assert fabric[3]["health"] == FAIL and fabric[3]["system"] == XGMI and fabric[3]["value"] == 6
    assert "xGMI link 4 (to GPU " in fabric[3]["message"] and "6 of 7 links up" in fabric[3]["message"]
    assert [g for g, v in fabric.items() if v["health"]] == [3] and doc["devices"][3]["xgmi_links_up"] == 6
    # everything else stays clean: a fault is charged to its GPU and its concern only
    assert [g for g, v in mem.items() if v["health"]] == [2, 6] and [g for g, v in power.items() if v["health"]] == [5]
    assert sorted(g for g, v in uniform.items() if v["health"]) == [1, 5]
    assert "GPU 6" in err and "FAIL" in err


def test_a_host_link_that_trained_low_is_found_from_sysfs(tmp_path):
    """The link state comes from the PCI device's sysfs files (what the kernel negotiated against what the card can do)."""
    rc, doc, _ = preflight("gpus=4")
    bdfs = [d["pci_bdf"] for d in doc["devices"]]
    assert not by_gpu(doc, "preflight_pcie")  # no sysfs entry for a scripted GPU: no verdict, not a guess
    for g, bdf in enumerate(bdfs):
        d = tmp_path / "bus" / "pci" / "devices" / bdf
        d.mkdir(parents=True)
        (d / "current_link_speed").write_text("16.0 GT/s PCIe\n" if g == 3 else "32.0 GT/s PCIe\n")
        (d / "current_link_width").write_text("8\n" if g == 3 else "16\n")
        (d / "max_link_speed").write_text("32.0 GT/s PCIe\n")
        (d / "max_link_width").write_text("16\n")
    (tmp_path / "module" / "amdgpu").mkdir(parents=True)
    (tmp_path / "module" / "amdgpu" / "version").write_text("6.16.6\n")
    rc, doc, err = preflight("gpus=4", env={"GPUMON_SYSFS_ROOT": str(tmp_path)})
    link = by_gpu(doc, "preflight_pcie")
    assert rc == 0 and doc["result"] == "warn" and doc["node"]["amdgpu_module"] == "6.16.6"
    assert sorted(link) == [0, 1, 2, 3] and [g for g, v in link.items() if v["health"]] == [3]
    assert link[3]["health"] == WARN and link[3]["system"] == PCIE and link[3]["value"] == 408
    assert "generation 4 x8" in link[3]["message"] and "generation 5 x16" in link[3]["message"]
    assert doc["devices"][3]["link_gen"] == 4 and doc["devices"][3]["link_width"] == 8 and doc["devices"][0]["link_gen"] == 5


def test_the_gpu_count_is_held_against_what_the_node_should_have():
    rc, doc, _ = preflight("gpus=7", "--expect-gpus", "8")
    v = by_gpu(doc, "preflight_runtime")[None]
    assert rc == 1 and v["health"] == FAIL and "7 GPU(s) enumerated, 8 expected" in v["message"]
    rc, doc, _ = preflight("gpus=8", "--expect-gpus", "8")
    assert rc == 0
    # one GPU: nothing to compare it with
    rc, doc, _ = preflight("gpus=1")
    assert rc == 0 and not by_gpu(doc, "preflight_uniform") and not by_gpu(doc, "preflight_fabric")
    # a library that cannot be opened is a finding, not a crash
    rc, doc, _ = preflight("gpus=8,fail_init=1")
    assert rc == 1 and by_gpu(doc, "preflight_runtime")[None]["health"] == FAIL


def test_the_report_reaches_the_exporter(tmp_path):
    rep = tmp_path / "preflight.json"
    e = dict(os.environ, GPUMON_BACKEND_SPEC="gpus=4,at_start=bad_pages+gpu=2+pending=1")
    e.pop("GPUMON_BACKEND", None)
    r = subprocess.run([DIAG, "preflight", "--backend", "fake", "--no-runtime", "--publish", str(rep)], capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode == 0 and "Result: WARN" in r.stdout, r.stdout[-300:] + r.stderr[-300:]
    out = tmp_path / "m.prom"
    r = subprocess.run([bin_path("gpumon-exporter"), "-e", "--once", "-o", str(out), "--backend", "fake", "--backend-spec", "gpus=4", "--diag-results", str(tmp_path)],
                       capture_output=True, text=True, timeout=60, env=e)
    assert r.returncode == 0, r.stderr[-400:]
    text = out.read_text()
    line = next(ln for ln in text.splitlines() if "preflight_memory" in ln and 'gpu="2"' in ln and ln.startswith("gpumon_diag_test_health"))
    assert line.rstrip().endswith(" 10")
