#include "views.h"

#include <cmath>
#include <cstring>
#include <ctime>
#include <map>
#include <mutex>
#include <unistd.h>

#include "../common/json.h"
#include "../common/util.h"

namespace gpumon {
namespace {

std::string opt_i(bool valid, int64_t v) { return valid ? strprintf("%lld", (long long)v) : "N/A"; }
std::string opt_d(bool valid, double v, const char* fmt = "%.3f") { return valid ? strprintf(fmt, v) : "N/A"; }
std::string opt_s(const char* s) { return s && *s ? s : "N/A"; }

void j_opt_i(JsonWriter& w, const char* k, bool valid, int64_t v) {
  w.key(k);
  if (valid) w.integer(v); else w.null();
}
void j_opt_d(JsonWriter& w, const char* k, bool valid, double v) {
  w.key(k);
  if (valid) w.num(v); else w.null();
}

// One persistent single-GPU group per (engine, gpu) for the health view.
std::mutex g_view_mu;
std::map<std::pair<Engine*, uint32_t>, gm_group_t> g_health_groups;

}  // namespace

std::string format_time_us(int64_t ts_us) {
  if (ts_us == 0) return "Running";
  time_t t = (time_t)(ts_us / 1000000);
  struct tm tm;
  localtime_r(&t, &tm);
  char buf[64];
  strftime(buf, sizeof buf, "%Y-%m-%d %H:%M:%S %z %Z", &tm);
  return buf;
}

gm_status_t view_device_info(Engine* e, uint32_t gpu, bool json, std::string* out) {
  gm_device_attributes_t a;
  a.version = GM_DEVICE_ATTRIBUTES_VERSION;
  gm_status_t st = e->device_attributes(gpu, &a);
  if (st != GM_ST_OK) return st;
  gm_topology_t topo;
  topo.version = GM_TOPOLOGY_VERSION;
  if (e->device_topology(gpu, &topo) != GM_ST_OK) topo.num_peers = 0;
  bool have_power = !std::isnan(a.power_cap_w);
  if (json) {
    JsonWriter w;
    w.begin_obj();
    w.key("GPU"); w.integer(a.gpu);
    w.key("Supported"); w.str(a.supported ? "Yes" : "No");
    w.key("DCGMSupported"); w.str(a.supported ? "Yes" : "No");
    w.key("UUID"); w.str(a.uuid);
    j_opt_i(w, "Power", have_power, (int64_t)llround(a.power_cap_w));
    w.key("PCI"); w.begin_obj();
    w.key("BusID"); w.str(a.pci_bdf);
    j_opt_i(w, "BAR1", a.vis_vram_total_mib >= 0, a.vis_vram_total_mib);
    j_opt_i(w, "FBTotal", a.vram_total_mib >= 0, a.vram_total_mib);
    j_opt_i(w, "Bandwidth", a.pcie_bandwidth_mbs >= 0, a.pcie_bandwidth_mbs);
    w.end_obj();
    w.key("Clocks"); w.begin_obj();
    j_opt_i(w, "Cores", a.gfx_clock_max_mhz >= 0, a.gfx_clock_max_mhz);
    j_opt_i(w, "Memory", a.mem_clock_max_mhz >= 0, a.mem_clock_max_mhz);
    w.end_obj();
    w.key("Identifiers"); w.begin_obj();
    w.key("Brand"); w.str(a.brand);
    w.key("Model"); w.str(a.name);
    w.key("Serial"); w.str(a.serial);
    w.key("Vbios"); w.str(a.vbios);
    w.key("InforomImageVersion"); w.str(a.firmware);
    w.key("FirmwareVersion"); w.str(a.firmware);
    w.key("DriverVersion"); w.str(a.driver_version);
    w.key("GfxTarget"); w.str(a.gfx_target);
    w.end_obj();
    w.key("Partition"); w.begin_obj();
    w.key("Compute"); w.str(a.compute_partition);
    w.key("Memory"); w.str(a.memory_partition);
    w.end_obj();
    w.key("Path"); w.str(a.device_path);
    w.key("XCDs"); w.integer(a.num_xcd);
    w.key("CUs"); w.integer(a.num_cu);
    w.key("Topology");
    if (!topo.num_peers) w.null();
    else {
      w.begin_arr();
      for (uint32_t i = 0; i < topo.num_peers; i++) {
        const gm_peer_link_t& p = topo.peers[i];
        w.begin_obj();
        w.key("GPU"); w.integer(p.gpu);
        w.key("BusID"); w.str(p.pci_bdf);
        w.key("Link"); w.str(link_type_name(p.link, p.hops));
        w.key("Hops"); w.integer(p.hops);
        w.key("Weight"); w.integer((long long)p.weight);
        j_opt_i(w, "MaxBandwidth", p.max_bw_mbs >= 0, p.max_bw_mbs);
        w.end_obj();
      }
      w.end_arr();
    }
    w.key("CPUAffinity"); w.str(a.cpu_affinity);
    w.end_obj();
    *out = w.text() + "\n";
    return GM_ST_OK;
  }
  std::string s;
  s += strprintf("Driver Version         : %s\n", opt_s(a.driver_version).c_str());
  s += strprintf("GPU                    : %u\n", a.gpu);
  s += strprintf("Supported              : %s\n", a.supported ? "Yes" : "No");
  s += strprintf("UUID                   : %s\n", a.uuid);
  s += strprintf("Brand                  : %s\n", opt_s(a.brand).c_str());
  s += strprintf("Model                  : %s\n", opt_s(a.name).c_str());
  s += strprintf("Serial Number          : %s\n", opt_s(a.serial).c_str());
  s += strprintf("Vbios                  : %s\n", opt_s(a.vbios).c_str());
  s += strprintf("Firmware Version       : %s\n", opt_s(a.firmware).c_str());
  s += strprintf("Gfx Target             : %s (%u XCDs, %u CUs)\n", opt_s(a.gfx_target).c_str(), a.num_xcd, a.num_cu);
  s += strprintf("Partition              : %s / %s\n", opt_s(a.compute_partition).c_str(), opt_s(a.memory_partition).c_str());
  s += strprintf("Bus ID                 : %s\n", a.pci_bdf);
  s += strprintf("Visible VRAM (MB)      : %s\n", opt_i(a.vis_vram_total_mib >= 0, a.vis_vram_total_mib).c_str());
  s += strprintf("HBM Memory (MB)        : %s\n", opt_i(a.vram_total_mib >= 0, a.vram_total_mib).c_str());
  s += strprintf("Bandwidth (MB/s)       : %s\n", opt_i(a.pcie_bandwidth_mbs >= 0, a.pcie_bandwidth_mbs).c_str());
  s += strprintf("Cores (MHz)            : %s\n", opt_i(a.gfx_clock_max_mhz >= 0, a.gfx_clock_max_mhz).c_str());
  s += strprintf("Memory (MHz)           : %s\n", opt_i(a.mem_clock_max_mhz >= 0, a.mem_clock_max_mhz).c_str());
  s += strprintf("Power (W)              : %s\n", opt_i(have_power, (int64_t)llround(a.power_cap_w)).c_str());
  s += strprintf("CPUAffinity            : %s\n", opt_s(a.cpu_affinity).c_str());
  s += "P2P Available          : ";
  if (!topo.num_peers) s += "None\n";
  else {
    for (uint32_t i = 0; i < topo.num_peers; i++)
      s += strprintf("\n    GPU%u - (BusID)%s - %s", topo.peers[i].gpu, topo.peers[i].pci_bdf,
                     link_type_name(topo.peers[i].link, topo.peers[i].hops));
    s += "\n";
  }
  s += "---------------------------------------------------------------------\n";
  *out = s;
  return GM_ST_OK;
}

gm_status_t view_device_status(Engine* e, uint32_t gpu, bool json, std::string* out) {
  gm_device_status_t d;
  d.version = GM_DEVICE_STATUS_VERSION;
  gm_status_t st = e->device_status(gpu, &d);
  if (st != GM_ST_OK) return st;
  auto ok = [&](int bit) { return (d.valid >> bit) & 1; };
  if (json) {
    JsonWriter w;
    w.begin_obj();
    j_opt_d(w, "Power", ok(GM_DS_POWER), d.power_w);
    j_opt_i(w, "Temperature", ok(GM_DS_TEMP), d.temperature_c);
    j_opt_i(w, "MemoryTemperature", ok(GM_DS_MEM_TEMP), d.mem_temp_c);
    w.key("Utilization"); w.begin_obj();
    j_opt_i(w, "GPU", ok(GM_DS_GFX_UTIL), d.gfx_util);
    j_opt_i(w, "Memory", ok(GM_DS_MEM_UTIL), d.mem_util);
    j_opt_i(w, "Encoder", ok(GM_DS_ENC_UTIL), d.enc_util);
    j_opt_i(w, "Decoder", ok(GM_DS_DEC_UTIL), d.dec_util);
    w.end_obj();
    w.key("Memory"); w.begin_obj();
    j_opt_i(w, "GlobalUsed", ok(GM_DS_VRAM_USED), d.vram_used_mib);
    j_opt_i(w, "GlobalFree", ok(GM_DS_VRAM_FREE), d.vram_free_mib);
    w.key("ECCErrors"); w.begin_obj();
    j_opt_i(w, "SingleBit", ok(GM_DS_ECC_CE), d.ecc_ce);
    j_opt_i(w, "DoubleBit", ok(GM_DS_ECC_UE), d.ecc_ue);
    w.end_obj();
    w.end_obj();
    w.key("Clocks"); w.begin_obj();
    j_opt_i(w, "Cores", ok(GM_DS_GFX_CLOCK), d.gfx_clock_mhz);
    j_opt_i(w, "Memory", ok(GM_DS_MEM_CLOCK), d.mem_clock_mhz);
    w.end_obj();
    w.key("PCI"); w.begin_obj();
    j_opt_i(w, "BAR1Used", ok(GM_DS_VIS_VRAM_USED), d.vis_vram_used_mib);
    w.key("Throughput"); w.begin_obj();
    j_opt_i(w, "Rx", ok(GM_DS_PCIE_RX), d.pcie_rx_kb / 1000);
    j_opt_i(w, "Tx", ok(GM_DS_PCIE_TX), d.pcie_tx_kb / 1000);
    j_opt_i(w, "Replays", ok(GM_DS_PCIE_REPLAYS), d.pcie_replays);
    w.end_obj();
    j_opt_i(w, "FBUsed", ok(GM_DS_VRAM_USED), d.vram_used_mib);
    w.end_obj();
    j_opt_i(w, "Performance", ok(GM_DS_PERF_LEVEL), d.perf_level);
    j_opt_i(w, "FanSpeed", ok(GM_DS_FAN), d.fan_speed);
    j_opt_i(w, "Throttle", ok(GM_DS_THROTTLE), d.throttle_status);
    w.end_obj();
    *out = w.text() + "\n";
    return GM_ST_OK;
  }
  std::string s;
  s += strprintf("Power (W)               : %s\n", opt_d(ok(GM_DS_POWER), d.power_w).c_str());
  s += strprintf("Temperature (°C)        : %s\n", opt_i(ok(GM_DS_TEMP), d.temperature_c).c_str());
  s += strprintf("Gfx Utilization (%%)     : %s\n", opt_i(ok(GM_DS_GFX_UTIL), d.gfx_util).c_str());
  s += strprintf("Memory Utilization (%%)  : %s\n", opt_i(ok(GM_DS_MEM_UTIL), d.mem_util).c_str());
  s += strprintf("Encoder Utilization (%%) : %s\n", opt_i(ok(GM_DS_ENC_UTIL), d.enc_util).c_str());
  s += strprintf("Decoder Utilization (%%) : %s\n", opt_i(ok(GM_DS_DEC_UTIL), d.dec_util).c_str());
  s += strprintf("Memory Clock (MHz)      : %s\n", opt_i(ok(GM_DS_MEM_CLOCK), d.mem_clock_mhz).c_str());
  s += strprintf("Gfx Clock (MHz)         : %s\n", opt_i(ok(GM_DS_GFX_CLOCK), d.gfx_clock_mhz).c_str());
  s += strprintf("HBM Used (MB)           : %s\n", opt_i(ok(GM_DS_VRAM_USED), d.vram_used_mib).c_str());
  *out = s;
  return GM_ST_OK;
}

gm_status_t view_health(Engine* e, uint32_t gpu, bool json, std::string* out) {
  gm_group_t grp = 0;
  std::vector<gm_health_response_t> resp;
  gm_status_t st = GM_ST_NOT_FOUND;
  for (int attempt = 0; attempt < 2 && st != GM_ST_OK; attempt++) {
    {
      std::lock_guard<std::mutex> l(g_view_mu);
      auto it = g_health_groups.find({e, gpu});
      if (it != g_health_groups.end()) grp = it->second;
      else {
        static int seq = 0;
        st = e->group_create(strprintf("health-view-%u-%d-%d", gpu, (int)getpid(), seq++), &grp);
        if (st != GM_ST_OK) return st;
        if ((st = e->group_add(grp, gpu)) != GM_ST_OK || (st = e->health_set(grp, GM_HEALTH_WATCH_ALL)) != GM_ST_OK) {
          e->group_destroy(grp);
          return st;
        }
        g_health_groups[{e, gpu}] = grp;
      }
    }
    st = e->health_check(grp, &resp);
    if (st == GM_ST_NOT_FOUND || st == GM_ST_NOT_CONFIGURED) {  // stale handle from an earlier engine
      std::lock_guard<std::mutex> l(g_view_mu);
      g_health_groups.erase({e, gpu});
    } else if (st != GM_ST_OK) {
      return st;
    }
  }
  if (st != GM_ST_OK || resp.empty()) return st == GM_ST_OK ? GM_ST_NO_DATA : st;
  const gm_health_response_t& r = resp[0];
  if (json) {
    JsonWriter w;
    w.begin_obj();
    w.key("GPU"); w.integer(r.gpu);
    w.key("Status"); w.str(health_result_name(r.overall));
    w.key("Watches"); w.begin_arr();
    for (uint32_t i = 0; i < r.incident_count; i++) {
      w.begin_obj();
      w.key("Type"); w.str(health_system_name(r.incidents[i].system));
      w.key("Status"); w.str(health_result_name(r.incidents[i].health));
      w.key("Error"); w.str(r.incidents[i].message);
      w.end_obj();
    }
    w.end_arr();
    w.end_obj();
    *out = w.text() + "\n";
    return GM_ST_OK;
  }
  std::string s = strprintf("GPU                : %u\nStatus             : %s\n", r.gpu, health_result_name(r.overall));
  for (uint32_t i = 0; i < r.incident_count; i++)
    s += strprintf("\nType               : %s\nStatus             : %s\nError              : %s\n",
                   health_system_name(r.incidents[i].system), health_result_name(r.incidents[i].health),
                   r.incidents[i].message);
  *out = s;
  return GM_ST_OK;
}

gm_status_t view_process_info(Engine* e, uint32_t pid, bool json, std::string* out) {
  std::unique_ptr<gm_pid_info_t> info(new gm_pid_info_t);
  info->version = GM_PID_INFO_VERSION;
  gm_status_t st = e->pid_info(GM_GROUP_ALL_GPUS, pid, info.get());
  if (st == GM_ST_NOT_WATCHED) {
    st = e->watch_pid_fields(GM_GROUP_ALL_GPUS, GM_DEFAULT_UPDATE_FREQ_US, GM_DEFAULT_MAX_KEEP_AGE_S, 0);
    if (st != GM_ST_OK) return st;
    st = e->pid_info(GM_GROUP_ALL_GPUS, pid, info.get());
  }
  if (st != GM_ST_OK) return st;
  std::string s;
  JsonWriter w;
  if (json) w.begin_arr();
  for (uint32_t i = 0; i < info->num_gpus; i++) {
    const gm_pid_gpu_stats_t& p = info->gpus[i];
    auto ok = [&](int bit) { return (p.valid >> bit) & 1; };
    if (json) {
      w.begin_obj();
      w.key("GPU"); w.integer(p.gpu);
      w.key("PID"); w.integer(p.pid);
      w.key("Name"); w.str(p.name);
      w.key("ProcessUtilization"); w.begin_obj();
      w.key("StartTime"); w.integer(p.start_ts_us / 1000000);
      w.key("EndTime"); w.integer(p.end_ts_us / 1000000);
      j_opt_i(w, "EnergyConsumed", ok(0), p.energy_mj / 1000);
      j_opt_d(w, "SmUtil", ok(3), p.proc_gfx_util_avg);
      j_opt_d(w, "MemUtil", ok(4), p.proc_mem_util_avg);
      w.end_obj();
      w.key("PCI"); w.begin_obj();
      w.key("Throughput"); w.begin_obj();
      j_opt_i(w, "Rx", ok(5), p.pcie_rx_avg_kbs / 1000);
      j_opt_i(w, "Tx", ok(6), p.pcie_tx_avg_kbs / 1000);
      j_opt_i(w, "Replays", ok(7), p.pcie_replays);
      w.end_obj();
      w.end_obj();
      w.key("Memory"); w.begin_obj();
      j_opt_i(w, "GlobalUsed", ok(8), p.max_vram_bytes);
      w.key("ECCErrors"); w.begin_obj();
      j_opt_i(w, "SingleBit", ok(9), p.ecc_ce);
      j_opt_i(w, "DoubleBit", ok(10), p.ecc_ue);
      w.end_obj();
      w.end_obj();
      w.key("GpuUtilization"); w.begin_obj();
      j_opt_d(w, "GPU", ok(1), p.gfx_util_avg);
      j_opt_d(w, "Memory", ok(2), p.mem_util_avg);
      w.end_obj();
      w.key("Clocks"); w.begin_obj();
      j_opt_i(w, "Cores", ok(11), p.gfx_clock_avg);
      j_opt_i(w, "Memory", ok(12), p.mem_clock_avg);
      w.end_obj();
      w.key("Violations"); w.begin_obj();
      j_opt_i(w, "Power", ok(13), p.power_violation_us);
      j_opt_i(w, "Thermal", ok(14), p.thermal_violation_us);
      j_opt_i(w, "Reliability", ok(15), p.reliability_violation_us);
      j_opt_i(w, "BoardLimit", ok(16), p.board_limit_violation_us);
      j_opt_i(w, "LowUtilization", ok(17), p.low_util_violation_us);
      j_opt_i(w, "SyncBoost", ok(18), p.sync_boost_violation_us);
      w.end_obj();
      w.key("XIDErrors"); w.begin_obj();
      w.key("NumErrors"); w.integer(p.num_critical_events);
      w.key("TimeStamp"); w.begin_arr();
      for (uint32_t k = 0; k < p.num_critical_events && k < GM_MAX_EVENT_TS; k++) w.integer(p.critical_event_ts_us[k] / 1000000);
      w.end_arr();
      w.end_obj();
      w.end_obj();
      continue;
    }
    s += "----------------------------------------------------------------------\n";
    s += strprintf("GPU ID                       : %u\n", p.gpu);
    s += "----------Execution Stats---------------------------------------------\n";
    s += strprintf("PID                          : %u\n", p.pid);
    s += strprintf("Name                         : %s\n", opt_s(p.name).c_str());
    s += strprintf("Start Time                   : %s\n", format_time_us(p.start_ts_us).c_str());
    s += strprintf("End Time                     : %s\n", format_time_us(p.end_ts_us).c_str());
    s += "----------Performance Stats-------------------------------------------\n";
    s += strprintf("Energy Consumed (Joules)     : %s\n", opt_i(ok(0), p.energy_mj / 1000).c_str());
    s += strprintf("Max GPU Memory Used (bytes)  : %s\n", opt_i(ok(8), p.max_vram_bytes).c_str());
    s += strprintf("Avg Gfx Clock (MHz)          : %s\n", opt_i(ok(11), p.gfx_clock_avg).c_str());
    s += strprintf("Avg Memory Clock (MHz)       : %s\n", opt_i(ok(12), p.mem_clock_avg).c_str());
    // Each label shows its own quantity (the reference swaps the two: restApi/handlers/utils.go:62-63).
    s += strprintf("Avg Gfx Utilization (%%)      : %s\n", opt_d(ok(1), p.gfx_util_avg, "%.1f").c_str());
    s += strprintf("Avg Memory Utilization (%%)   : %s\n", opt_d(ok(2), p.mem_util_avg, "%.1f").c_str());
    s += strprintf("Avg PCIe Rx Bandwidth (MB)   : %s\n", opt_i(ok(5), p.pcie_rx_avg_kbs / 1000).c_str());
    s += strprintf("Avg PCIe Tx Bandwidth (MB)   : %s\n", opt_i(ok(6), p.pcie_tx_avg_kbs / 1000).c_str());
    s += "----------Event Stats-------------------------------------------------\n";
    s += strprintf("Correctable ECC Errors       : %s\n", opt_i(ok(9), p.ecc_ce).c_str());
    s += strprintf("Uncorrectable ECC Errors     : %s\n", opt_i(ok(10), p.ecc_ue).c_str());
    s += strprintf("Critical GPU Events          : %u\n", p.num_critical_events);
    s += "----------Slowdown Stats----------------------------------------------\n";
    s += strprintf("Due to - Power (us)          : %s\n", opt_i(ok(13), p.power_violation_us).c_str());
    s += strprintf("       - Thermal (us)        : %s\n", opt_i(ok(14), p.thermal_violation_us).c_str());
    s += strprintf("       - Reliability (us)    : %s\n", opt_i(ok(15), p.reliability_violation_us).c_str());
    s += strprintf("       - Board Limit (us)    : %s\n", opt_i(ok(16), p.board_limit_violation_us).c_str());
    s += strprintf("       - Low Utilization (us): %s\n", opt_i(ok(17), p.low_util_violation_us).c_str());
    s += strprintf("       - Sync Boost (us)     : %s\n", opt_i(ok(18), p.sync_boost_violation_us).c_str());
    s += "----------Process Utilization-----------------------------------------\n";
    s += strprintf("Avg Gfx Utilization (%%)      : %s\n", opt_d(ok(3), p.proc_gfx_util_avg, "%.1f").c_str());
    s += strprintf("Avg Memory Utilization (%%)   : %s\n", opt_d(ok(4), p.proc_mem_util_avg, "%.1f").c_str());
    s += "----------------------------------------------------------------------\n";
  }
  if (json) {
    w.end_arr();
    *out = w.text() + "\n";
  } else {
    *out = s;
  }
  return GM_ST_OK;
}

gm_status_t view_hostengine(Engine* e, bool json, std::string* out) {
  gm_introspect_t in;
  in.version = GM_INTROSPECT_VERSION;
  gm_status_t st = e->introspect(&in);
  if (st == GM_ST_NOT_CONFIGURED) {
    e->introspect_toggle(true);
    st = e->introspect(&in);
  }
  if (st != GM_ST_OK) return st;
  if (json) {
    JsonWriter w;
    w.begin_obj();
    w.key("Memory"); w.integer(in.memory_kb);
    w.key("CPU"); w.num(in.cpu_percent);
    w.key("SampleCycles"); w.integer((long long)in.sample_cycles);
    w.key("LastCycleMs"); w.num(in.last_cycle_ms);
    w.key("AvgCycleMs"); w.num(in.avg_cycle_ms);
    w.key("BackendErrors"); w.integer((long long)in.backend_errors);
    w.end_obj();
    *out = w.text() + "\n";
  } else {
    *out = strprintf("Memory(KB)      : %lld\nCPU(%%)          : %.2f\n", (long long)in.memory_kb, in.cpu_percent);
  }
  return GM_ST_OK;
}

gm_status_t view_topology_matrix(Engine* e, std::string* out) {
  uint32_t n = 0;
  gm_status_t st = e->device_count(&n);
  if (st != GM_ST_OK) return st;
  std::vector<gm_topology_t> topo(n);
  for (uint32_t i = 0; i < n; i++) {
    topo[i].version = GM_TOPOLOGY_VERSION;
    if ((st = e->device_topology(i, &topo[i])) != GM_ST_OK) return st;
  }
  std::string s = "      ";
  for (uint32_t i = 0; i < n; i++) s += strprintf("GPU%-4u", i);
  s += "CPUAffinity\n";
  for (uint32_t i = 0; i < n; i++) {
    s += strprintf("GPU%-3u", i);
    for (uint32_t j = 0; j < n; j++) {
      const char* cell = "N/A";
      if (i == j) cell = "X";
      else
        for (uint32_t k = 0; k < topo[i].num_peers; k++)
          if (topo[i].peers[k].gpu == j) cell = link_type_name(topo[i].peers[k].link, topo[i].peers[k].hops);
      s += strprintf("%-7s", cell);
    }
    s += strprintf("%s\n", topo[i].cpu_affinity);
  }
  s += "\nLegend:\n X     = Self\n SYS   = Connection traversing PCIe as well as the interconnect between CPU sockets\n"
       " NODE  = Connection traversing PCIe as well as the interconnect between PCIe Host Bridges within a NUMA node\n"
       " PHB   = Connection traversing PCIe as well as a PCIe Host Bridge (typically the CPU)\n"
       " PXB   = Connection traversing multiple PCIe switches (without traversing the PCIe Host Bridge)\n"
       " PIX   = Connection traversing a single PCIe switch\n"
       " PSB   = Connection traversing a single on-board PCIe switch\n"
       " XGMI# = Connection over xGMI (Infinity Fabric) in # hops\n";
  *out = s;
  return GM_ST_OK;
}

gm_status_t view_links(Engine* e, uint32_t gpu, bool json, std::string* out) {
  std::vector<gm_xgmi_link_t> links;
  gm_status_t st = e->xgmi_links(gpu, &links);
  if (st != GM_ST_OK) return st;
  const int n = (int)links.size();
  auto opt = [](int64_t v) { return v < 0 ? std::string("N/A") : std::to_string(v); };
  auto jopt = [](int64_t v) { return v < 0 ? std::string("null") : std::to_string(v); };
  std::string s;
  if (json) {
    s = "{\"gpu\":" + std::to_string(gpu) + ",\"links\":[";
    for (int i = 0; i < n; i++) {
      const gm_xgmi_link_t& l = links[i];
      s += std::string(i ? "," : "") + "{\"link\":" + std::to_string(l.link) + ",\"state\":" +
           (l.state < 0 ? "null" : l.state ? "\"up\"" : "\"down\"") + ",\"peer_gpu\":" + jopt(l.peer_gpu) + ",\"peer_bdf\":" +
           (l.peer_bdf[0] ? "\"" + std::string(l.peer_bdf) + "\"" : std::string("null")) + ",\"bit_rate_gbps\":" + jopt(l.bit_rate_gbps) +
           ",\"max_bandwidth_gbps\":" + jopt(l.max_bandwidth_gbps) + ",\"rx_kb\":" + jopt(l.rx_kb) + ",\"tx_kb\":" + jopt(l.tx_kb) + "}";
    }
    s += "]}\n";
  } else {
    s = strprintf("xGMI links of GPU %u\n%-5s %-6s %-9s %-14s %10s %12s %16s %16s\n", gpu, "Link", "State", "Peer GPU", "Peer bus id", "Gb/s/lane",
                  "Max Gb/s", "RX KB", "TX KB");
    for (int i = 0; i < n; i++) {
      const gm_xgmi_link_t& l = links[i];
      s += strprintf("%-5u %-6s %-9s %-14s %10s %12s %16s %16s\n", l.link, l.state < 0 ? "N/A" : l.state ? "Up" : "Down",
                     opt(l.peer_gpu).c_str(), l.peer_bdf[0] ? l.peer_bdf : "N/A", opt(l.bit_rate_gbps).c_str(),
                     opt(l.max_bandwidth_gbps).c_str(), opt(l.rx_kb).c_str(), opt(l.tx_kb).c_str());
    }
    if (!n) s += "(no xGMI link reported)\n";
  }
  *out = s;
  return GM_ST_OK;
}

gm_status_t view_diag_results(Engine* e, uint32_t gpu, bool json, std::string* out) {
  std::vector<gm_diag_result_t> res;
  gm_status_t st = e->diag_results(gpu, &res);
  if (st != GM_ST_OK) return st;
  const int n = (int)res.size();
  std::string s;
  if (json) {
    s = "{\"gpu\":" + std::to_string(gpu) + ",\"results\":[";
    for (int i = 0; i < n; i++) {
      const gm_diag_result_t& r = res[i];
      std::string msg;
      for (const char* c = r.message; *c; c++) {
        if (*c == '"' || *c == '\\') msg += '\\';
        msg += (unsigned char)*c < 0x20 ? ' ' : *c;
      }
      std::string test;
      for (const char* c = r.test; *c; c++) {
        if (*c == '"' || *c == '\\') test += '\\';
        test += (unsigned char)*c < 0x20 ? ' ' : *c;
      }
      s += std::string(i ? "," : "") + "{\"test\":\"" + test + "\",\"system\":\"" + health_system_name(r.system) + "\",\"health\":\"" +
           health_result_name(r.health) + "\",\"value\":" + (std::isnan(r.value) ? std::string("null") : strprintf("%.6g", r.value)) +
           ",\"time\":\"" + format_time_us(r.ts_us) + "\",\"message\":\"" + msg + "\"}";
    }
    s += "]}\n";
  } else {
    s = strprintf("Diagnostics of GPU %u\n%-14s %-10s %-9s %12s  %-24s %s\n", gpu, "Test", "System", "Result", "Value", "When", "Message");
    for (int i = 0; i < n; i++) {
      const gm_diag_result_t& r = res[i];
      s += strprintf("%-14s %-10s %-9s %12s  %-24s %s\n", r.test, health_system_name(r.system), health_result_name(r.health),
                     std::isnan(r.value) ? "N/A" : strprintf("%.6g", r.value).c_str(), format_time_us(r.ts_us).c_str(), r.message);
    }
    if (!n) s += "(no diagnostic has reported)\n";
  }
  *out = s;
  return GM_ST_OK;
}

std::string view_violation(const gm_violation_t& v) {
  std::string s = strprintf("GPU %u: %s at %s: ", v.gpu, policy_condition_name(v.condition), format_time_us(v.ts_us).c_str());
  switch (v.condition) {
    case GM_POLICY_COND_ECC_UE: s += strprintf("{Location:%s NumErrors:%u}", v.d.ecc.location, v.d.ecc.count); break;
    case GM_POLICY_COND_PCIE: s += strprintf("{ReplayCounter:%u}", v.d.pcie.replay_counter); break;
    case GM_POLICY_COND_MAX_BAD_PAGES: s += strprintf("{CePages:%u UePages:%u}", v.d.bad_pages.ce_pages, v.d.bad_pages.ue_pages); break;
    case GM_POLICY_COND_THERMAL: s += strprintf("{ThermalViolation:%u}", v.d.thermal.temperature_c); break;
    case GM_POLICY_COND_POWER: s += strprintf("{PowerViolation:%u}", v.d.power.power_w); break;
    case GM_POLICY_COND_XGMI: s += strprintf("{FieldId:%u Link:%d Counter:%u}", v.d.xgmi.field_id, v.d.xgmi.link == 0xffff ? -1 : (int)v.d.xgmi.link, v.d.xgmi.counter); break;
    case GM_POLICY_COND_GPU_EVENT: s += strprintf("{ErrNum:%u %s}", v.d.event.code, v.d.event.message); break;
  }
  return s;
}

}  // namespace gpumon
