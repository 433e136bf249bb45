#include "metrics_table.h"

#include <cstring>

#include "../common/util.h"

namespace gpumon {
namespace {

// Byte offsets of format 1 / content 8, established against the device library's decoding of the same bytes
// (scripts/probes/metrics_layout_probe.cc, scripts/metrics_layout_match.py) on an idle and on a loaded GPU.
constexpr size_t kSize = 3872;
constexpr size_t kTempHotspot = 4, kTempMem = 6, kTempVrSoc = 8, kSocketPower = 10, kGfxActivity = 12, kUmcActivity = 14;
constexpr size_t kEnergy = 24, kAccCounter = 40, kProchot = 44, kPpt = 48, kSocketThm = 52, kVrThm = 56, kHbmThm = 60;
constexpr size_t kPcieWidth = 68, kPcieSpeed = 70, kXgmiWidth = 72, kXgmiSpeed = 74;
constexpr size_t kPcieBwInst = 96, kPcieL0Recov = 104, kPcieReplay = 112, kNakSent = 128, kNakRcvd = 132;
constexpr size_t kXgmiRead = 136, kXgmiWrite = 200, kXgmiStatus = 264, kFwTimestamp = 288;
constexpr size_t kGfxClk = 296, kSocClk = 312, kVclk = 320, kDclk = 328, kUclk = 336, kNumPartition = 338;
constexpr size_t kXcp0 = 344, kXcpStride = 440;
// inside one partition's block: 8 x u32 busy, 40 x u16 jpeg, 4 x u16 vcn, then five arrays of 8 x u64
constexpr size_t kXcpBusyInst = 0, kXcpLowUtil = 312, kXcpHostLimitTotal = 376;
static_assert(kXcp0 + kMtXcp * kXcpStride + 8 == kSize, "partition blocks end eight bytes before the table does");

template <typename T>
T rd(const uint8_t* raw, size_t off) {
  T v;
  memcpy(&v, raw + off, sizeof v);  // the table is little-endian, like every host this runs on
  return v;
}
// A narrower counter into a wider member: all-ones stays all-ones.
uint64_t widen32(uint32_t v) { return v == 0xffffffffu ? ~0ull : v; }

}  // namespace

void metrics_table_clear(MetricsTable* m) { memset(m, 0xff, sizeof *m); }

bool metrics_table_decode(const uint8_t* raw, size_t n, MetricsTable* out, std::string* why) {
  auto no = [&](const std::string& s) {
    if (why) *why = s;
    return false;
  };
  if (!raw || n < 4) return no("short read");
  const uint16_t size = rd<uint16_t>(raw, 0);
  const unsigned format = raw[2], content = raw[3];
  if (format != 1 || content != 8) return no(strprintf("table format %u.%u (this decoder reads 1.8)", format, content));
  if (size != kSize || n < kSize) return no(strprintf("table of %u bytes, %zu read (1.8 has %zu)", (unsigned)size, n, kSize));
  metrics_table_clear(out);
  out->temperature_hotspot = rd<uint16_t>(raw, kTempHotspot);
  out->temperature_mem = rd<uint16_t>(raw, kTempMem);
  out->temperature_vrsoc = rd<uint16_t>(raw, kTempVrSoc);
  out->current_socket_power = rd<uint16_t>(raw, kSocketPower);
  out->average_gfx_activity = rd<uint16_t>(raw, kGfxActivity);
  out->average_umc_activity = rd<uint16_t>(raw, kUmcActivity);
  out->energy_accumulator = rd<uint64_t>(raw, kEnergy);
  out->accumulation_counter = widen32(rd<uint32_t>(raw, kAccCounter));
  out->prochot_residency_acc = widen32(rd<uint32_t>(raw, kProchot));
  out->ppt_residency_acc = widen32(rd<uint32_t>(raw, kPpt));
  out->socket_thm_residency_acc = widen32(rd<uint32_t>(raw, kSocketThm));
  out->vr_thm_residency_acc = widen32(rd<uint32_t>(raw, kVrThm));
  out->hbm_thm_residency_acc = widen32(rd<uint32_t>(raw, kHbmThm));
  out->pcie_link_width = rd<uint16_t>(raw, kPcieWidth);
  out->pcie_link_speed = rd<uint16_t>(raw, kPcieSpeed);
  out->xgmi_link_width = rd<uint16_t>(raw, kXgmiWidth);
  out->xgmi_link_speed = rd<uint16_t>(raw, kXgmiSpeed);
  out->pcie_bandwidth_inst = rd<uint64_t>(raw, kPcieBwInst);
  out->pcie_l0_to_recov_count_acc = rd<uint64_t>(raw, kPcieL0Recov);
  out->pcie_replay_count_acc = rd<uint64_t>(raw, kPcieReplay);
  out->pcie_nak_sent_count_acc = rd<uint32_t>(raw, kNakSent);
  out->pcie_nak_rcvd_count_acc = rd<uint32_t>(raw, kNakRcvd);
  for (int i = 0; i < kMtLinks; i++) {
    out->xgmi_read_data_acc[i] = rd<uint64_t>(raw, kXgmiRead + 8 * i);
    out->xgmi_write_data_acc[i] = rd<uint64_t>(raw, kXgmiWrite + 8 * i);
    out->xgmi_link_status[i] = rd<uint16_t>(raw, kXgmiStatus + 2 * i);
  }
  out->firmware_timestamp = rd<uint64_t>(raw, kFwTimestamp);
  for (int i = 0; i < kMtGfxClks; i++) out->current_gfxclks[i] = rd<uint16_t>(raw, kGfxClk + 2 * i);
  for (int i = 0; i < kMtClks; i++) {
    out->current_socclks[i] = rd<uint16_t>(raw, kSocClk + 2 * i);
    out->current_vclk0s[i] = rd<uint16_t>(raw, kVclk + 2 * i);
    out->current_dclk0s[i] = rd<uint16_t>(raw, kDclk + 2 * i);
  }
  out->current_uclk = rd<uint16_t>(raw, kUclk);
  // the single-valued members of the older layouts carry the first instance
  out->current_gfxclk = out->current_gfxclks[0];
  out->current_socclk = out->current_socclks[0];
  out->current_vclk0 = out->current_vclk0s[0];
  out->current_dclk0 = out->current_dclk0s[0];
  out->num_partition = rd<uint16_t>(raw, kNumPartition);
  for (int p = 0; p < kMtXcp; p++) {
    const size_t base = kXcp0 + p * kXcpStride;
    for (int x = 0; x < kMtXcc; x++) {
      out->xcp_stats[p].gfx_busy_inst[x] = rd<uint32_t>(raw, base + kXcpBusyInst + 4 * x);
      out->xcp_stats[p].gfx_low_utilization_acc[x] = rd<uint64_t>(raw, base + kXcpLowUtil + 8 * x);
      out->xcp_stats[p].gfx_below_host_limit_total_acc[x] = rd<uint64_t>(raw, base + kXcpHostLimitTotal + 8 * x);
    }
  }
  return true;
}

// One list of members for the comparison and for the JSON, so that neither can forget one.
#define MT_SCALARS(X)                                                                                                  \
  X(temperature_edge) X(temperature_hotspot) X(temperature_mem) X(temperature_vrgfx) X(temperature_vrsoc) X(temperature_vrmem) \
  X(average_gfx_activity) X(average_umc_activity) X(average_mm_activity) X(average_socket_power) X(current_socket_power)  \
  X(current_gfxclk) X(current_socclk) X(current_uclk) X(current_vclk0) X(current_dclk0) X(pcie_link_width) X(pcie_link_speed) \
  X(xgmi_link_width) X(xgmi_link_speed) X(num_partition) X(pcie_nak_sent_count_acc) X(pcie_nak_rcvd_count_acc)           \
  X(energy_accumulator) X(firmware_timestamp) X(pcie_bandwidth_inst) X(pcie_l0_to_recov_count_acc) X(pcie_replay_count_acc) \
  X(accumulation_counter) X(prochot_residency_acc) X(ppt_residency_acc) X(socket_thm_residency_acc) X(vr_thm_residency_acc) \
  X(hbm_thm_residency_acc)
#define MT_ARRAYS(X)                                                                                                   \
  X(temperature_hbm, kMtHbm) X(current_gfxclks, kMtGfxClks) X(current_socclks, kMtClks) X(current_vclk0s, kMtClks)       \
  X(current_dclk0s, kMtClks) X(vcn_activity, kMtVcn) X(xgmi_link_status, kMtLinks) X(xgmi_read_data_acc, kMtLinks)       \
  X(xgmi_write_data_acc, kMtLinks)
#define MT_XCP_ARRAYS(X) X(gfx_busy_inst) X(gfx_below_host_limit_acc) X(gfx_below_host_limit_total_acc) X(gfx_low_utilization_acc)

std::string metrics_table_diff(const MetricsTable& a, const MetricsTable& b) {
#define X(name) \
  if (a.name != b.name) return strprintf(#name ": %llu against %llu", (unsigned long long)a.name, (unsigned long long)b.name);
  MT_SCALARS(X)
#undef X
#define X(name, n)                                                                                                      \
  for (int i = 0; i < n; i++)                                                                                           \
    if (a.name[i] != b.name[i])                                                                                         \
      return strprintf(#name "[%d]: %llu against %llu", i, (unsigned long long)a.name[i], (unsigned long long)b.name[i]);
  MT_ARRAYS(X)
#undef X
  for (int p = 0; p < kMtXcp; p++)
    for (int x = 0; x < kMtXcc; x++) {
#define X(name)                                                                                                         \
  if (a.xcp_stats[p].name[x] != b.xcp_stats[p].name[x])                                                                 \
    return strprintf("xcp_stats[%d]." #name "[%d]: %llu against %llu", p, x, (unsigned long long)a.xcp_stats[p].name[x],  \
                     (unsigned long long)b.xcp_stats[p].name[x]);
      MT_XCP_ARRAYS(X)
#undef X
    }
  return "";
}

std::string metrics_table_json(const MetricsTable& m) {
  std::string j = "{";
#define X(name) j += strprintf("\"" #name "\":%llu,", (unsigned long long)m.name);
  MT_SCALARS(X)
#undef X
#define X(name, n)                                                                      \
  j += "\"" #name "\":[";                                                               \
  for (int i = 0; i < n; i++) j += strprintf("%s%llu", i ? "," : "", (unsigned long long)m.name[i]); \
  j += "],";
  MT_ARRAYS(X)
#undef X
  j += "\"xcp_stats\":[";
  for (int p = 0; p < kMtXcp; p++) {
    j += p ? ",{" : "{";
    bool first = true;
#define X(name)                                                                          \
  j += std::string(first ? "" : ",") + "\"" #name "\":[";                                \
  first = false;                                                                         \
  for (int x = 0; x < kMtXcc; x++) j += strprintf("%s%llu", x ? "," : "", (unsigned long long)m.xcp_stats[p].name[x]); \
  j += "]";
    MT_XCP_ARRAYS(X)
#undef X
    j += "}";
  }
  j += "]}";
  return j;
}

}  // namespace gpumon
