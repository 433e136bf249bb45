"""The driver's bulk metrics table decoded without the device library (core/backend/metrics_table.cc).

The fixture holds raw tables of an MI355X (format 1, content 8) next to what the device library made of the very same
bytes: the decoder must agree on every member the sampler uses.  The hardware test in tests/test_gpu_telemetry.py
checks the running host the same way."""
import json
import os

import pytest

from k8s_gpu_monitor_amd import engine as E

FIXTURE = os.path.join(os.path.dirname(__file__), "data", "gpu_metrics_v1_8.json")
U16, U32, U64 = 0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF


@pytest.fixture(scope="module")
def pairs():
    with open(FIXTURE) as f:
        return json.load(f)["pairs"]


def test_decoding_matches_the_device_library_member_by_member(pairs):
    scalars = ["temperature_edge", "temperature_hotspot", "temperature_mem", "temperature_vrgfx", "temperature_vrsoc",
               "temperature_vrmem", "average_gfx_activity", "average_umc_activity", "average_mm_activity", "average_socket_power",
               "current_socket_power", "current_gfxclk", "current_socclk", "current_uclk", "current_vclk0", "current_dclk0",
               "pcie_link_width", "pcie_link_speed", "xgmi_link_width", "xgmi_link_speed", "num_partition",
               "pcie_nak_sent_count_acc", "pcie_nak_rcvd_count_acc", "energy_accumulator", "firmware_timestamp",
               "pcie_bandwidth_inst", "pcie_l0_to_recov_count_acc", "pcie_replay_count_acc", "accumulation_counter",
               "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc"]
    arrays = ["temperature_hbm", "current_gfxclks", "current_socclks", "current_vclk0s", "current_dclk0s", "vcn_activity",
              "xgmi_link_status", "xgmi_read_data_acc", "xgmi_write_data_acc"]
    per_xcp = ["gfx_busy_inst", "gfx_below_host_limit_acc", "gfx_below_host_limit_total_acc", "gfx_low_utilization_acc"]
    for pair in pairs:
        mine = E.metrics_table_decode(bytes.fromhex(pair["raw"]))
        for k in scalars + arrays:
            assert mine[k] == pair[k], k
        for p in range(8):
            for k in per_xcp:
                assert mine["xcp_stats"][p][k] == pair[f"xcp{p}"][k], (p, k)
    # the pairs are not three copies of one table: the loaded GPU draws its power cap and runs its clocks up
    idle, load = E.metrics_table_decode(bytes.fromhex(pairs[0]["raw"])), E.metrics_table_decode(bytes.fromhex(pairs[-1]["raw"]))
    assert idle["current_socket_power"] < 400 < 1300 < load["current_socket_power"]
    assert idle["current_gfxclks"][0] < 200 < 1500 < load["current_gfxclks"][0]
    assert load["average_gfx_activity"] == 100 and load["xcp_stats"][0]["gfx_busy_inst"] == [100] * 8
    assert load["xcp_stats"][1]["gfx_busy_inst"] == [U32] * 8  # one partition: the others are not reported
    assert load["xgmi_link_status"] == [U16] + [1] * 7 and load["temperature_edge"] == U16 and load["vcn_activity"] == [U16] * 4


def test_other_layouts_are_refused_with_a_reason(pairs):
    raw = bytearray(bytes.fromhex(pairs[0]["raw"]))
    with pytest.raises(ValueError, match=r"format 1\.7"):
        E.metrics_table_decode(bytes(raw[:2]) + bytes([1, 7]) + bytes(raw[4:]))
    with pytest.raises(ValueError, match="3872"):
        E.metrics_table_decode(bytes(raw[:-8]))  # a truncated read
    with pytest.raises(ValueError, match="3864 bytes"):
        E.metrics_table_decode((3864).to_bytes(2, "little") + bytes(raw[2:]))  # a header that announces another size
    with pytest.raises(ValueError, match="short"):
        E.metrics_table_decode(b"\x01")


def test_narrow_counters_keep_their_not_reported_marker(pairs):
    raw = bytearray(bytes.fromhex(pairs[0]["raw"]))
    raw[48:52] = b"\xff" * 4  # the 32-bit power-limit residency: not reported
    m = E.metrics_table_decode(bytes(raw))
    assert m["ppt_residency_acc"] == U64 and m["accumulation_counter"] == pairs[0]["accumulation_counter"]


def test_other_backends_have_no_table(engine):
    assert E.metrics_table_source(0) == "n/a"


def test_any_bytes_decode_or_are_refused_never_crash(pairs):
    """Whatever the driver's file holds, the decoder answers with a table or a reason."""
    from hypothesis import given, settings, strategies as st
    good = bytes.fromhex(pairs[0]["raw"])

    @settings(max_examples=200, deadline=None)
    @given(st.binary(min_size=0, max_size=4200), st.integers(0, 3871), st.binary(min_size=1, max_size=64))
    def run(blob, at, patch):
        for raw in (blob, good[:at] + patch + good[at + len(patch):]):
            try:
                m = E.metrics_table_decode(raw)
            except ValueError as e:
                assert str(e)
                continue
            assert len(m["xcp_stats"]) == 8 and len(m["current_gfxclks"]) == 8
            assert all(0 <= v <= U64 for v in m["xgmi_read_data_acc"])

    run()
