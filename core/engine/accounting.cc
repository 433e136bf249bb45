// Per-process accounting.  Mirrors the reference's WatchPidFields / GetPidInfo
// pair (bindings/go/dcgm/process_info.go:64-189): start recording on a group
// of GPUs, later ask for one PID and get, per GPU it ran on, start / end time,
// energy, average utilisation and clocks, PCIe traffic, peak memory, ECC
// deltas, throttle time and critical-event timestamps.
//
// On every accounting tick the engine lists the processes of each watched GPU
// (KFD process table through the backend) and integrates the device-level
// telemetry of the snapshot cache over the lifetime of each process.  Energy is
// attributed by the process' share of the gfx engine time when the driver
// reports it, else split evenly between the processes present.
#include <cstring>

#include "../common/util.h"
#include "engine.h"

namespace gpumon {
namespace {

const uint16_t kAcctFields[] = {GM_FI_POWER_USAGE, GM_FI_GFX_UTIL, GM_FI_MEM_UTIL, GM_FI_GFX_CLOCK,
                                GM_FI_MEM_CLOCK, GM_FI_PCIE_RX_KB, GM_FI_PCIE_TX_KB, GM_FI_PCIE_REPLAY_COUNTER,
                                GM_FI_ECC_CE_VOLATILE, GM_FI_ECC_UE_VOLATILE, GM_FI_POWER_VIOLATION,
                                GM_FI_THERMAL_VIOLATION, GM_FI_RELIABILITY_VIOLATION, GM_FI_BOARD_LIMIT_VIOLATION,
                                GM_FI_LOW_UTIL_VIOLATION, GM_FI_VRAM_USED};
const uint16_t kViolFields[5] = {GM_FI_POWER_VIOLATION, GM_FI_THERMAL_VIOLATION, GM_FI_RELIABILITY_VIOLATION,
                                 GM_FI_BOARD_LIMIT_VIOLATION, GM_FI_LOW_UTIL_VIOLATION};

}  // namespace

gm_status_t LocalEngine::watch_pid_fields(gm_group_t g, int64_t freq_us, double keep_age_s, int32_t keep_samples) {
  (void)keep_samples;
  if (freq_us < 1000 || keep_age_s < 0) return GM_ST_BADPARAM;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    std::vector<uint32_t> gp;
    gm_status_t st = resolve_group(g, &gp);
    if (st != GM_ST_OK) return st;
    if (gp.empty()) return GM_ST_GROUP_INCOMPATIBLE;
    for (uint32_t gpu : gp) {
      PidWatch& pw = pid_watch_[gpu];
      pw.freq_us = pw.freq_us ? std::min(pw.freq_us, freq_us) : freq_us;
      pw.keep_age_us = std::max(pw.keep_age_us, (int64_t)(keep_age_s * 1e6));
    }
    const size_t n = sizeof kAcctFields / sizeof kAcctFields[0];
    ensure_internal_watch("accounting", gp, std::vector<uint16_t>(kAcctFields, kAcctFields + n), freq_us);
  }
  return update_all(true);
}

void LocalEngine::accounting_tick(int64_t now_mono, int64_t now_wall) {
  std::vector<uint32_t> watched;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    for (auto& kv : pid_watch_) watched.push_back(kv.first);
  }
  for (uint32_t gpu : watched) {
    std::vector<ProcSample> ps;
    if (backend_->processes(gpu, &ps) != GM_ST_OK) {
      backend_errors_++;
      continue;
    }
    double power = 0, gfx = 0, mem = 0, gclk = 0, mclk = 0;
    bool have_power = get_d(gpu, GM_FI_POWER_USAGE, &power);
    bool have_gfx = get_d(gpu, GM_FI_GFX_UTIL, &gfx);
    bool have_mem = get_d(gpu, GM_FI_MEM_UTIL, &mem);
    bool have_gclk = get_d(gpu, GM_FI_GFX_CLOCK, &gclk);
    bool have_mclk = get_d(gpu, GM_FI_MEM_CLOCK, &mclk);
    int64_t rx = -1, tx = -1, rp = -1, ce = -1, ue = -1, viol[5];
    if (!get_i(gpu, GM_FI_PCIE_RX_KB, &rx)) rx = -1;
    if (!get_i(gpu, GM_FI_PCIE_TX_KB, &tx)) tx = -1;
    if (!get_i(gpu, GM_FI_PCIE_REPLAY_COUNTER, &rp)) rp = -1;
    if (!get_i(gpu, GM_FI_ECC_CE_VOLATILE, &ce)) ce = -1;
    if (!get_i(gpu, GM_FI_ECC_UE_VOLATILE, &ue)) ue = -1;
    for (int i = 0; i < 5; i++)
      if (!get_i(gpu, kViolFields[i], &viol[i])) viol[i] = -1;
    std::vector<int64_t> events;
    {
      std::lock_guard<std::mutex> l(gpus_[gpu]->hist_mu);
      events = gpus_[gpu]->critical_event_ts;
    }

    std::lock_guard<std::mutex> l(state_mu_);
    auto pit = pid_watch_.find(gpu);
    if (pit == pid_watch_.end()) continue;
    PidWatch& pw = pit->second;
    double dt_s = pw.last_tick_mono ? (double)(now_mono - pw.last_tick_mono) / 1e6 : 0;
    pw.last_tick_mono = now_mono;

    // Engine-time deltas decide the energy split.
    std::map<uint32_t, double> share;
    double total_delta = 0;
    for (auto& p : ps) {
      auto it = pw.procs.find(p.pid);
      double d = 0;
      if (it != pw.procs.end() && it->second.have_engine && p.gfx_engine_ns >= it->second.last_engine_ns)
        d = (double)(p.gfx_engine_ns - it->second.last_engine_ns);
      share[p.pid] = d;
      total_delta += d;
    }
    for (auto& p : ps) share[p.pid] = total_delta > 0 ? share[p.pid] / total_delta : 1.0 / (double)ps.size();

    for (auto& p : ps) {
      ProcAcct& a = pw.procs[p.pid];
      if (a.pid == 0 || a.end_us != 0) {  // new process (or a reused pid)
        a = ProcAcct();
        a.pid = p.pid;
        a.name = p.name.empty() ? process_name(p.pid) : p.name;
        a.start_us = now_wall;
        a.pcie_rx0 = rx; a.pcie_tx0 = tx; a.replay0 = rp; a.ce0 = ce; a.ue0 = ue;
        for (int i = 0; i < 5; i++) a.viol0[i] = viol[i];
      }
      a.last_seen_mono = now_mono;
      if (p.vram_bytes != GM_PROC_MEM_UNKNOWN) a.max_vram = std::max(a.max_vram, p.vram_bytes), a.have_vram = true;
      a.last_engine_ns = p.gfx_engine_ns;
      a.have_engine = p.gfx_engine_ns != 0;
      double sh = share[p.pid];
      if (have_power && dt_s > 0) a.energy_mj += power * dt_s * 1000.0 * sh;
      if (have_gfx) a.sum_gfx += gfx;
      if (have_mem) a.sum_mem += mem;
      if (have_gclk) a.sum_gfxclk += gclk;
      if (have_mclk) a.sum_memclk += mclk;
      a.sum_share += sh;
      a.nsamples++;
      a.pcie_rx1 = rx; a.pcie_tx1 = tx; a.replay1 = rp; a.ce1 = ce; a.ue1 = ue;
      for (int i = 0; i < 5; i++) a.viol1[i] = viol[i];
      a.event_ts.clear();
      for (int64_t t : events)
        if (t >= a.start_us && a.event_ts.size() < GM_MAX_EVENT_TS) a.event_ts.push_back(t);
    }
    // Processes that vanished end now; ended records age out.
    for (auto it = pw.procs.begin(); it != pw.procs.end();) {
      ProcAcct& a = it->second;
      bool present = false;
      for (auto& p : ps) present = present || p.pid == a.pid;
      if (!present && a.end_us == 0) a.end_us = now_wall;
      if (a.end_us && pw.keep_age_us > 0 && now_wall - a.end_us > pw.keep_age_us) it = pw.procs.erase(it);
      else ++it;
    }
    while (pw.procs.size() > GM_MAX_PROCS) {
      auto oldest = pw.procs.end();
      for (auto it = pw.procs.begin(); it != pw.procs.end(); ++it)
        if (it->second.end_us && (oldest == pw.procs.end() || it->second.end_us < oldest->second.end_us)) oldest = it;
      if (oldest == pw.procs.end()) break;
      pw.procs.erase(oldest);
    }
  }
}

gm_status_t LocalEngine::pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out) {
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  out->pid = pid;
  std::lock_guard<std::mutex> l(state_mu_);
  std::vector<uint32_t> gp;
  gm_status_t st = resolve_group(g, &gp);
  if (st != GM_ST_OK) return st;
  bool watched = false;
  for (uint32_t gpu : gp) {
    auto pit = pid_watch_.find(gpu);
    if (pit == pid_watch_.end()) continue;
    watched = true;
    auto it = pit->second.procs.find(pid);
    if (it == pit->second.procs.end()) continue;
    const ProcAcct& a = it->second;
    gm_pid_gpu_stats_t& s = out->gpus[out->num_gpus++];
    s.gpu = gpu;
    s.pid = pid;
    s.start_ts_us = a.start_us;
    s.end_ts_us = a.end_us;
    copy_str(s.name, a.name);
    auto set = [&](int bit) { s.valid |= 1ull << bit; };
    double n = (double)a.nsamples;
    s.energy_mj = (int64_t)a.energy_mj; set(0);
    if (n > 0) {
      s.gfx_util_avg = a.sum_gfx / n; set(1);
      s.mem_util_avg = a.sum_mem / n; set(2);
      double share = a.sum_share / n;
      s.proc_gfx_util_avg = s.gfx_util_avg * share; set(3);
      s.proc_mem_util_avg = s.mem_util_avg * share; set(4);
      s.gfx_clock_avg = (int64_t)(a.sum_gfxclk / n); set(11);
      s.mem_clock_avg = (int64_t)(a.sum_memclk / n); set(12);
    }
    double life_s = (double)((a.end_us ? a.end_us : now_us()) - a.start_us) / 1e6;
    if (a.pcie_rx0 >= 0 && a.pcie_rx1 >= 0 && life_s > 0) { s.pcie_rx_avg_kbs = (int64_t)((double)(a.pcie_rx1 - a.pcie_rx0) / life_s); set(5); }
    if (a.pcie_tx0 >= 0 && a.pcie_tx1 >= 0 && life_s > 0) { s.pcie_tx_avg_kbs = (int64_t)((double)(a.pcie_tx1 - a.pcie_tx0) / life_s); set(6); }
    if (a.replay0 >= 0 && a.replay1 >= 0) { s.pcie_replays = a.replay1 - a.replay0; set(7); }
    if (a.have_vram) { s.max_vram_bytes = (int64_t)a.max_vram; set(8); }
    if (a.ce0 >= 0 && a.ce1 >= 0) { s.ecc_ce = a.ce1 - a.ce0; set(9); }
    if (a.ue0 >= 0 && a.ue1 >= 0) { s.ecc_ue = a.ue1 - a.ue0; set(10); }
    int64_t* vout[5] = {&s.power_violation_us, &s.thermal_violation_us, &s.reliability_violation_us,
                        &s.board_limit_violation_us, &s.low_util_violation_us};
    for (int i = 0; i < 5; i++)
      if (a.viol0[i] >= 0 && a.viol1[i] >= 0) { *vout[i] = a.viol1[i] - a.viol0[i]; set(13 + i); }
    s.num_critical_events = (uint32_t)a.event_ts.size();
    for (size_t i = 0; i < a.event_ts.size() && i < GM_MAX_EVENT_TS; i++) s.critical_event_ts_us[i] = a.event_ts[i];
    if (out->num_gpus >= GM_MAX_GPUS) break;
  }
  if (!watched) return GM_ST_NOT_WATCHED;
  if (!out->num_gpus) return GM_ST_NO_DATA;
  return GM_ST_OK;
}

}  // namespace gpumon
