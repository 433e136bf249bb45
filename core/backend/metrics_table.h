// The amdgpu driver's bulk metrics table, read and decoded without the device library.
//
// One read of /sys/bus/pci/devices/<bdf>/gpu_metrics is one copy of the firmware's table out of the driver's
// 1 ms cache: 16 us on an MI355X host.  The device library's amdsmi_get_gpu_metrics_info wraps the same read in
// 140 us of its own (locks, stream I/O, translation into its all-versions struct), which at 8 GPUs x 10 Hz is the
// whole CPU budget of the exporter.  This decoder knows the one layout MI355X produces today (format 1, content 8,
// 3872 bytes); the backend switches to it only after it has reproduced the library's decoding of the same bytes
// field for field on the running host, checks again every few minutes, and falls back to the library on any doubt.
#pragma once

#include <cstddef>
#include <cstdint>
#include <string>

namespace gpumon {

constexpr int kMtXcp = 8, kMtXcc = 8, kMtHbm = 4, kMtVcn = 4, kMtGfxClks = 8, kMtClks = 4, kMtLinks = 8;

// What the sampler consumes of the table, under the device library's member names.  All-ones = not reported.
struct MetricsTable {
  uint16_t temperature_edge, temperature_hotspot, temperature_mem, temperature_vrgfx, temperature_vrsoc, temperature_vrmem;
  uint16_t temperature_hbm[kMtHbm];
  uint16_t average_gfx_activity, average_umc_activity, average_mm_activity;
  uint16_t average_socket_power, current_socket_power;
  uint16_t current_gfxclk, current_socclk, current_uclk, current_vclk0, current_dclk0;
  uint16_t current_gfxclks[kMtGfxClks], current_socclks[kMtClks], current_vclk0s[kMtClks], current_dclk0s[kMtClks];
  uint16_t vcn_activity[kMtVcn];
  uint16_t pcie_link_width, pcie_link_speed, xgmi_link_width, xgmi_link_speed;
  uint16_t xgmi_link_status[kMtLinks];
  uint16_t num_partition;
  uint32_t pcie_nak_sent_count_acc, pcie_nak_rcvd_count_acc;
  uint64_t energy_accumulator, firmware_timestamp;
  uint64_t pcie_bandwidth_inst, pcie_l0_to_recov_count_acc, pcie_replay_count_acc;
  uint64_t xgmi_read_data_acc[kMtLinks], xgmi_write_data_acc[kMtLinks];
  uint64_t accumulation_counter, prochot_residency_acc, ppt_residency_acc, socket_thm_residency_acc, vr_thm_residency_acc,
      hbm_thm_residency_acc;
  struct Xcp {
    uint32_t gfx_busy_inst[kMtXcc];
    uint64_t gfx_below_host_limit_acc[kMtXcc], gfx_below_host_limit_total_acc[kMtXcc], gfx_low_utilization_acc[kMtXcc];
  } xcp_stats[kMtXcp];
};

void metrics_table_clear(MetricsTable* m);  // every member "not reported"
// Decodes the raw table; false (and why) for anything but the layout this file knows.
bool metrics_table_decode(const uint8_t* raw, size_t n, MetricsTable* out, std::string* why);
// First member that differs between two decodings of the same table, "" if none.
std::string metrics_table_diff(const MetricsTable& a, const MetricsTable& b);
std::string metrics_table_json(const MetricsTable& m);

}  // namespace gpumon
