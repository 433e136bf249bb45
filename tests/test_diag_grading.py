"""Verdict arithmetic of the active diagnostics, on synthetic measurements (no GPU): the expectation table derived
from the link rate, the reading of the three link matrices into findings that name GPU and peer, the comparison of a
node's GPUs with each other, and the three outcomes of the reference's health check
(/root/reference/bindings/go/dcgm/health.go:88-98: Healthy / Warning / Failure) as 0 / 10 / 20."""
import subprocess

import pytest

from conftest import bin_path

diag = pytest.importorskip("k8s_gpu_monitor_amd.ops.diag")


def _matrix(n, value, **cells):
    m = [[0.0 if i == j else float(value) for j in range(n)] for i in range(n)]
    for key, v in cells.items():
        i, j = (int(x) for x in key[1:].split("_"))
        m[i][j] = v
    return m


def test_expectations_follow_from_the_link_rate():
    # MI355X: 153.6 GB/s per link counts both directions; one direction is 76.8
    assert diag.link_expected_gbps("uni") == pytest.approx(76.8 * 0.70)
    assert diag.link_expected_gbps("bidi") == pytest.approx(76.8 * 0.65)
    assert diag.link_expected_gbps("all", peers=7) == pytest.approx(7 * 76.8 * 0.60)
    assert diag.link_expected_gbps("uni", pull=1) == pytest.approx(76.8 * 0.70 * 0.80)
    assert diag.link_expected_gbps("uni", link_gbps=64.0, derate_uni=0.5, scale=2.0) == pytest.approx(64.0)
    # collectives: a ring is bound by one link; the sharded ones can use all n-1 links of a GPU
    assert diag.rccl_expected_busbw("all_reduce", 1) == 0
    for n in (2, 4, 8):
        assert diag.rccl_expected_busbw("all_reduce", n) == pytest.approx(76.8 * 0.70)
        assert diag.rccl_expected_busbw("broadcast", n) == pytest.approx(76.8 * 0.70)
        for op in ("all_gather", "reduce_scatter", "all_to_all"):
            assert diag.rccl_expected_busbw(op, n) == pytest.approx(76.8 * (n - 1) * 0.50)
    assert diag.rccl_expected_busbw("all_to_all", 8, link_gbps=50.0, derate_collective=0.4) == pytest.approx(140.0)


def test_three_outcomes_against_an_expectation():
    assert diag.grade_against(100, 100) == 0
    assert diag.grade_against(90, 100) == 0
    assert diag.grade_against(89.9, 100) == 10
    assert diag.grade_against(75, 100) == 10
    assert diag.grade_against(74.9, 100) == 20
    assert diag.grade_against(float("nan"), 100) == 20
    assert diag.grade_against(5, 0) == 0          # no expectation: nothing to fail


def test_outlier_grading_of_the_gpus_of_a_node():
    worst, h, med = diag.outlier_grade([1600, 1610, 1590, 1605, 1500, 1598, 1602, 1400])
    assert h == [0, 0, 0, 0, 10, 0, 0, 20] and worst == 20
    assert med[7] == pytest.approx(1600.0)        # the median of the OTHER seven
    assert diag.outlier_grade([1600.0]) == (0, [0], [0.0])
    # a GPU that delivered nothing is the outlier, not the reference for the others
    worst, h, med = diag.outlier_grade([6.5, 6.6, float("nan"), 6.4])
    assert h[2] == 20 and h[0] == h[1] == h[3] == 0


def _kinds(findings):
    return sorted((f["kind"], f["src"], f["dst"], f["health"]) for f in findings)


def test_a_healthy_mesh_has_no_findings():
    worst, f = diag.link_diagnose(_matrix(8, 58.0), _matrix(8, 55.0), _matrix(8, 50.0))
    assert worst == 0 and f == []
    worst, f = diag.link_diagnose(_matrix(8, 47.0), None, _matrix(8, 40.0), pull=1)
    assert worst == 0 and f == []


def test_one_slow_link_one_half_duplex_link_one_gpu_with_low_egress_give_exactly_those_findings():
    """An 8-GPU mesh with three independent faults.  Link 2->5 is slow in every pattern; link 1<->6 carries its rate in
    one direction at a time only; GPU 4 cannot keep seven links busy at once.  Each fault is one finding (two for the
    two directions of the half-duplex link), named by GPU and peer, and none of them spills over to the neighbours."""
    uni = _matrix(8, 58.0, c2_5=30.0)
    bidi = _matrix(8, 55.0, c2_5=30.0, c1_6=29.0, c6_1=30.0)
    all_ = _matrix(8, 50.0, c2_5=28.0, c1_6=27.0, c6_1=27.0)
    for j in range(8):
        if j != 4:
            all_[4][j] = 33.0
    worst, f = diag.link_diagnose(uni, bidi, all_)
    assert worst == 20
    assert _kinds(f) == [("half_duplex", 1, 6, 20), ("half_duplex", 6, 1, 20), ("low_egress", 4, -1, 20),
                         ("slow_link", 2, 5, 20)]
    by = {f_["kind"] + str(f_["src"]): f_ for f_ in f}
    assert by["slow_link2"]["pattern"] == "uni" and "52 % of the other links' median 58.0" in by["slow_link2"]["message"]
    assert by["half_duplex1"]["pattern"] == "bidi" and "50 % of the 58.0 GB/s it carries alone" in by["half_duplex1"]["message"]
    assert by["low_egress4"]["pattern"] == "all" and "writes to 7 peers at once at 231.0 GB/s in total, 66 %" in by["low_egress4"]["message"]
    assert by["low_egress4"]["gbps"] == pytest.approx(231.0) and by["low_egress4"]["expected_gbps"] == pytest.approx(350.0)


def test_a_slow_receiver_and_a_pair_that_only_slows_under_load():
    all_ = _matrix(8, 50.0, c0_7=41.0)
    for i in range(8):
        if i != 3:
            all_[i][3] = 30.0
    worst, f = diag.link_diagnose(_matrix(8, 58.0), _matrix(8, 55.0), all_)
    assert _kinds(f) == [("low_ingress", -1, 3, 20), ("slow_under_load", 0, 7, 10)]
    assert "is written by 7 peers at once" in f[0]["message"] or "is written by 7 peers at once" in f[1]["message"]


def test_absolute_expectations_catch_a_uniformly_slow_fabric_and_the_scale_moves_them():
    # every link at 35 GB/s: nothing stands out, but one direction of a 76.8 GB/s link should carry ~54
    worst, f = diag.link_diagnose(_matrix(4, 35.0))
    assert worst == 20 and len(f) == 12 and all(x["kind"] == "slow_link" and "expected of one direction" in x["message"] for x in f)
    worst, f = diag.link_diagnose(_matrix(4, 45.0))
    assert worst == 10 and all(x["health"] == 10 for x in f)
    # a site that knows its links do less relaxes the table as a whole
    assert diag.link_diagnose(_matrix(4, 35.0), scale=0.7)[0] == 0
    # and the advertised rate of the engine's link table replaces the MI355X figure
    assert diag.link_diagnose(_matrix(4, 35.0), link_gbps=50.0)[0] == 0


def test_no_access_and_corrupt_data_fail_and_stay_out_of_the_statistics():
    uni = _matrix(4, 58.0, c2_3=-1.0)
    bad = [[0] * 4 for _ in range(4)]
    bad[1][0] = 7
    all_ = _matrix(4, 50.0, c2_3=-1.0, c1_0=5.0)
    worst, f = diag.link_diagnose(uni, None, all_, bad=bad)
    assert _kinds(f) == [("corrupt_data", 1, 0, 20), ("no_access", 2, 3, 20)]
    # one GPU: nothing to read, not an error
    assert diag.link_diagnose([[0.0]]) == (0, [])


def _diag(*args):
    return subprocess.run([bin_path("gpumon-diag"), *args], capture_output=True, text=True, timeout=60)


def test_expect_flags_are_checked_before_a_device_is_needed():
    r = _diag("hbm", "--expect", "hbm.copy")
    assert r.returncode == 2
    # the malformed item is named once a device is there; without one the run stops at the device
    assert "no GPU" in r.stderr or "cannot read 'hbm.copy'" in r.stderr or "no HIP device" in r.stderr
    r = _diag("gemm", "--gpus", "0,0")
    assert r.returncode == 2
