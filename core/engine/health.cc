// Health module.  Same contract as the reference's health path — enable watch
// systems on a group, then ask for an overall tri-state result plus a list of
// per-subsystem incidents (bindings/go/dcgm/health.go:26-124) — with AMD
// subsystems: PCIe, xGMI, HBM memory (RAS + bad pages), compute (XCD / GFX RAS),
// thermal, power, driver events, firmware.
//
// Counter-type checks report what happened since the previous check of the same
// (group, GPU); level-type checks (temperature, link state) report the present.
#include <cstring>

#include "../common/util.h"
#include "engine.h"

namespace gpumon {
namespace {

const uint16_t kHealthFields[] = {
    GM_FI_PCIE_REPLAY_COUNTER, GM_FI_PCIE_NAK_SENT, GM_FI_PCIE_NAK_RCVD, GM_FI_PCIE_L0_TO_RECOVERY,
    GM_FI_DEV_PCIE_GEN, GM_FI_DEV_PCIE_WIDTH,
    GM_FI_XGMI_LINKS_DOWN, GM_FI_XGMI_CRC_ERRORS, GM_FI_XGMI_DATA_ERRORS, GM_FI_XGMI_ERROR_STATUS,
    GM_FI_XGMI_LINK_STATUS_0, GM_FI_XGMI_LINK_STATUS_0 + 1, GM_FI_XGMI_LINK_STATUS_0 + 2, GM_FI_XGMI_LINK_STATUS_0 + 3,
    GM_FI_XGMI_LINK_STATUS_0 + 4, GM_FI_XGMI_LINK_STATUS_0 + 5, GM_FI_XGMI_LINK_STATUS_0 + 6, GM_FI_XGMI_LINK_STATUS_0 + 7,
    GM_FI_ECC_UMC_CE, GM_FI_ECC_UMC_UE, GM_FI_ECC_GFX_CE, GM_FI_ECC_GFX_UE, GM_FI_ECC_SDMA_UE, GM_FI_ECC_MMHUB_UE,
    GM_FI_ECC_CE_VOLATILE, GM_FI_ECC_UE_VOLATILE, GM_FI_ECC_DEFERRED,
    GM_FI_BAD_PAGES_CE, GM_FI_BAD_PAGES_UE, GM_FI_BAD_PAGES_PENDING,
    GM_FI_GPU_TEMP, GM_FI_MEM_TEMP, GM_FI_THERMAL_VIOLATION, GM_FI_POWER_VIOLATION, GM_FI_BOARD_LIMIT_VIOLATION,
    GM_FI_POWER_USAGE, GM_FI_DEV_POWER_CAP, GM_FI_THROTTLE_STATUS};

// Level thresholds for MI355X-class parts (degrees C, counts); gm_health_thresholds_set replaces them.
gm_health_thresholds_t default_thresholds() {
  gm_health_thresholds_t t;
  memset(&t, 0, sizeof t);
  t.version = GM_HEALTH_THRESHOLDS_VERSION;
  t.hotspot_warn_c = 100;
  t.hotspot_fail_c = 110;
  t.mem_warn_c = 95;
  t.mem_fail_c = 105;
  t.bad_pages_fail = 256;  // the RAS bad-page table fills up around this many entries
  t.ce_burst_warn = 100;   // correctable errors between two checks
  t.power_over_cap_warn = 1.10;
  return t;
}

}  // namespace

gm_health_thresholds_t health_default_thresholds() { return default_thresholds(); }

gm_status_t LocalEngine::health_thresholds_set(const gm_health_thresholds_t& t) {
  if (t.version != GM_HEALTH_THRESHOLDS_VERSION) return GM_ST_VER_MISMATCH;
  if (t.hotspot_warn_c <= 0 || t.hotspot_fail_c < t.hotspot_warn_c || t.mem_warn_c <= 0 || t.mem_fail_c < t.mem_warn_c ||
      t.bad_pages_fail <= 0 || t.ce_burst_warn <= 0 || !(t.power_over_cap_warn >= 1.0))
    return GM_ST_BADPARAM;
  std::lock_guard<std::mutex> l(state_mu_);
  thresholds_ = t;
  return GM_ST_OK;
}

gm_status_t LocalEngine::health_thresholds_get(gm_health_thresholds_t* t) {
  if (t->version != GM_HEALTH_THRESHOLDS_VERSION) return GM_ST_VER_MISMATCH;
  std::lock_guard<std::mutex> l(state_mu_);
  *t = thresholds_;
  return GM_ST_OK;
}

gm_status_t LocalEngine::health_set(gm_group_t g, uint32_t systems) {
  if (systems & ~GM_HEALTH_WATCH_ALL) return GM_ST_BADPARAM;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    std::vector<uint32_t> gp;
    gm_status_t st = resolve_group(g, &gp);
    if (st != GM_ST_OK) return st;
    groups_[g].health_systems = systems;  // creates the entry of the implicit all-GPUs group on first use
    if (systems) {
      const size_t n = sizeof kHealthFields / sizeof kHealthFields[0];
      ensure_internal_watch("health", gp, std::vector<uint16_t>(kHealthFields, kHealthFields + n),
                            GM_DEFAULT_UPDATE_FREQ_US);
    }
    for (uint32_t gpu : gp) health_base_[{g, gpu}] = HealthBaseline();
  }
  if (systems) return update_all(true);
  return GM_ST_OK;
}

gm_status_t LocalEngine::health_get(gm_group_t g, uint32_t* systems) {
  std::lock_guard<std::mutex> l(state_mu_);
  auto it = groups_.find(g);
  if (it == groups_.end()) return g == GM_GROUP_ALL_GPUS ? (*systems = 0, GM_ST_OK) : GM_ST_NOT_FOUND;
  *systems = it->second.health_systems;
  return GM_ST_OK;
}

gm_status_t LocalEngine::health_check(gm_group_t g, std::vector<gm_health_response_t>* out) {
  out->clear();
  std::vector<uint32_t> gp;
  uint32_t systems = 0;
  gm_health_thresholds_t th;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    gm_status_t st = resolve_group(g, &gp);
    if (st != GM_ST_OK) return st;
    auto it = groups_.find(g);
    if (it != groups_.end()) systems = it->second.health_systems;
    th = thresholds_;
  }
  const int64_t kHotspotWarn = th.hotspot_warn_c, kHotspotFail = th.hotspot_fail_c, kMemWarn = th.mem_warn_c, kMemFail = th.mem_fail_c;
  const int64_t kBadPagesFail = th.bad_pages_fail, kCeBurstWarn = th.ce_burst_warn;
  if (!systems) return GM_ST_NOT_CONFIGURED;

  for (uint32_t gpu : gp) {
    gm_health_response_t r;
    memset(&r, 0, sizeof r);
    r.version = GM_HEALTH_RESPONSE_VERSION;
    r.gpu = gpu;
    r.overall = GM_HEALTH_PASS;
    auto incident = [&](uint32_t system, int32_t health, const std::string& msg) {
      if (!(systems & system)) return;
      if (r.incident_count < GM_MAX_INCIDENTS) {
        gm_incident_t& in = r.incidents[r.incident_count++];
        in.system = system;
        in.health = health;
        copy_str(in.message, msg);
      }
      if (health > r.overall) r.overall = health;
    };

    HealthBaseline base;
    {
      std::lock_guard<std::mutex> l(state_mu_);
      base = health_base_[{g, gpu}];
    }
    HealthBaseline next = base;
    next.valid = true;
    next.last_check_us = now_us();
    // Increase of a counter since the previous check; the first check after
    // health_set only records the baseline.
    auto delta = [&](uint16_t id) -> int64_t {
      int64_t cur;
      if (!get_i(gpu, id, &cur)) return 0;
      next.counters[id] = cur;
      auto it = base.counters.find(id);
      if (!base.valid || it == base.counters.end()) return 0;
      return cur > it->second ? cur - it->second : 0;
    };
    int64_t v, w;

    // ---- PCIe
    if ((v = delta(GM_FI_PCIE_REPLAY_COUNTER)) > 0)
      incident(GM_HEALTH_WATCH_PCIE, GM_HEALTH_WARN, strprintf("Detected %lld PCIe replays since the last check.", (long long)v));
    if ((v = delta(GM_FI_PCIE_NAK_SENT) + delta(GM_FI_PCIE_NAK_RCVD)) > 0)
      incident(GM_HEALTH_WATCH_PCIE, GM_HEALTH_WARN, strprintf("Detected %lld PCIe NAKs since the last check.", (long long)v));
    if ((v = delta(GM_FI_PCIE_L0_TO_RECOVERY)) > 0)
      incident(GM_HEALTH_WATCH_PCIE, GM_HEALTH_WARN, strprintf("PCIe link entered recovery %lld times since the last check.", (long long)v));
    if (get_i(gpu, GM_FI_DEV_PCIE_GEN, &v) && get_i(gpu, GM_FI_DEV_PCIE_WIDTH, &w)) {
      const GpuStatic& st = gpus_[gpu]->st;
      if (st.pcie_max_gen && st.pcie_max_width && v > 0 && w > 0 && (v < st.pcie_max_gen || w < st.pcie_max_width))
        incident(GM_HEALTH_WATCH_PCIE, GM_HEALTH_WARN,
                 strprintf("PCIe link trained at Gen%lld x%lld, device supports Gen%u x%u.", (long long)v, (long long)w,
                           st.pcie_max_gen, st.pcie_max_width));
    }

    // ---- xGMI
    for (int l = 0; l < GM_MAX_XGMI_LINKS; l++)
      if (get_i(gpu, GM_FI_XGMI_LINK_STATUS_0 + l, &v) && v == 0)
        incident(GM_HEALTH_WATCH_XGMI, GM_HEALTH_FAIL, strprintf("xGMI link %d is down.", l));
    if (get_i(gpu, GM_FI_XGMI_ERROR_STATUS, &v) && v > 0)
      incident(GM_HEALTH_WATCH_XGMI, GM_HEALTH_WARN, v > 1 ? "The xGMI link layer has latched several errors." : "The xGMI link layer has latched an error.");
    if ((v = delta(GM_FI_XGMI_DATA_ERRORS)) > 0)
      incident(GM_HEALTH_WATCH_XGMI, GM_HEALTH_FAIL, strprintf("Detected %lld uncorrectable xGMI errors since the last check.", (long long)v));
    if ((v = delta(GM_FI_XGMI_CRC_ERRORS)) > 0)
      incident(GM_HEALTH_WATCH_XGMI, GM_HEALTH_WARN, strprintf("Detected %lld correctable xGMI link errors since the last check.", (long long)v));

    // ---- HBM memory
    if ((v = delta(GM_FI_ECC_UMC_UE)) > 0)
      incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_FAIL, strprintf("Detected %lld uncorrectable HBM (UMC) ECC errors since the last check.", (long long)v));
    if ((v = delta(GM_FI_ECC_DEFERRED)) > 0)
      incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_WARN, strprintf("Detected %lld deferred (poison) errors since the last check.", (long long)v));
    if ((v = delta(GM_FI_ECC_UMC_CE)) >= kCeBurstWarn)
      incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_WARN, strprintf("Detected a burst of %lld correctable HBM ECC errors since the last check.", (long long)v));
    if (get_i(gpu, GM_FI_BAD_PAGES_PENDING, &v) && v > 0)
      incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_WARN, strprintf("%lld HBM pages are pending retirement; reset the GPU to retire them.", (long long)v));
    {
      int64_t ce = 0, ue = 0;
      bool have = get_i(gpu, GM_FI_BAD_PAGES_CE, &ce);
      have = get_i(gpu, GM_FI_BAD_PAGES_UE, &ue) || have;
      if (have && ce + ue >= kBadPagesFail)
        incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_FAIL, strprintf("%lld HBM pages are retired; the bad-page table is nearly full.", (long long)(ce + ue)));
      else if ((v = delta(GM_FI_BAD_PAGES_UE)) > 0)
        incident(GM_HEALTH_WATCH_MEM, GM_HEALTH_WARN, strprintf("%lld HBM pages were retired for uncorrectable errors since the last check.", (long long)v));
      else
        delta(GM_FI_BAD_PAGES_UE);
    }

    // ---- compute
    if ((v = delta(GM_FI_ECC_GFX_UE)) > 0)
      incident(GM_HEALTH_WATCH_COMPUTE, GM_HEALTH_FAIL, strprintf("Detected %lld uncorrectable errors in the GFX (XCD) block since the last check.", (long long)v));
    if ((v = delta(GM_FI_ECC_SDMA_UE) + delta(GM_FI_ECC_MMHUB_UE)) > 0)
      incident(GM_HEALTH_WATCH_COMPUTE, GM_HEALTH_FAIL, strprintf("Detected %lld uncorrectable errors in the SDMA/MMHUB blocks since the last check.", (long long)v));
    if ((v = delta(GM_FI_ECC_GFX_CE)) >= kCeBurstWarn)
      incident(GM_HEALTH_WATCH_COMPUTE, GM_HEALTH_WARN, strprintf("Detected a burst of %lld correctable errors in the GFX (XCD) block since the last check.", (long long)v));

    // ---- thermal
    if (get_i(gpu, GM_FI_GPU_TEMP, &v)) {
      if (v >= kHotspotFail) incident(GM_HEALTH_WATCH_THERMAL, GM_HEALTH_FAIL, strprintf("Hotspot temperature %lld C is at or above the %lld C shutdown region.", (long long)v, (long long)kHotspotFail));
      else if (v >= kHotspotWarn) incident(GM_HEALTH_WATCH_THERMAL, GM_HEALTH_WARN, strprintf("Hotspot temperature %lld C is at or above %lld C.", (long long)v, (long long)kHotspotWarn));
    }
    if (get_i(gpu, GM_FI_MEM_TEMP, &v)) {
      if (v >= kMemFail) incident(GM_HEALTH_WATCH_THERMAL, GM_HEALTH_FAIL, strprintf("HBM temperature %lld C is at or above %lld C.", (long long)v, (long long)kMemFail));
      else if (v >= kMemWarn) incident(GM_HEALTH_WATCH_THERMAL, GM_HEALTH_WARN, strprintf("HBM temperature %lld C is at or above %lld C.", (long long)v, (long long)kMemWarn));
    }
    if ((v = delta(GM_FI_THERMAL_VIOLATION)) > 0)
      incident(GM_HEALTH_WATCH_THERMAL, GM_HEALTH_WARN, strprintf("Clocks were thermally throttled for %lld us since the last check.", (long long)v));

    // ---- power
    if ((v = delta(GM_FI_POWER_VIOLATION)) > 0)
      incident(GM_HEALTH_WATCH_POWER, GM_HEALTH_WARN, strprintf("Clocks were power-throttled for %lld us since the last check.", (long long)v));
    if ((v = delta(GM_FI_BOARD_LIMIT_VIOLATION)) > 0)
      incident(GM_HEALTH_WATCH_POWER, GM_HEALTH_WARN, strprintf("PROCHOT / VR limits throttled clocks for %lld us since the last check.", (long long)v));
    {
      double p, cap;
      if (get_d(gpu, GM_FI_POWER_USAGE, &p) && get_d(gpu, GM_FI_DEV_POWER_CAP, &cap) && cap > 0 && p > cap * th.power_over_cap_warn)
        incident(GM_HEALTH_WATCH_POWER, GM_HEALTH_WARN,
                 strprintf("Socket power %.0f W exceeds the %.0f W cap by more than %.0f%%.", p, cap, (th.power_over_cap_warn - 1.0) * 100.0));
    }

    // ---- driver / RAS events
    {
      GpuState& gs = *gpus_[gpu];
      std::lock_guard<std::mutex> l(gs.hist_mu);
      size_t total = gs.critical_event_ts.size();
      size_t fresh = base.valid && total >= base.events_seen ? total - base.events_seen : 0;
      next.events_seen = total;
      if (fresh) {
        int32_t sev = (gs.last_event_code == GM_EVENT_VMFAULT) ? GM_HEALTH_WARN : GM_HEALTH_FAIL;
        incident(GM_HEALTH_WATCH_DRIVER, sev,
                 strprintf("%zu critical driver event(s) since the last check; last: %s.", fresh, event_code_name(gs.last_event_code)));
      }
    }
    {
      uint64_t errs = backend_errors_;
      if (base.valid && errs > base.backend_errors)
        incident(GM_HEALTH_WATCH_DRIVER, GM_HEALTH_WARN,
                 strprintf("%llu device-library calls failed since the last check.", (unsigned long long)(errs - base.backend_errors)));
      next.backend_errors = errs;
    }

    // ---- active diagnostics: the latest result of every test that did not pass stays an incident of its system
    {
      GpuState& gs = *gpus_[gpu];
      std::lock_guard<std::mutex> l(gs.hist_mu);
      for (auto& kv : gs.diag)
        if (kv.second.health != GM_HEALTH_PASS)
          incident(kv.second.system, kv.second.health, strprintf("Diagnostic %s: %s", kv.second.test, kv.second.message));
    }

    // ---- firmware
    if (gpus_[gpu]->st.firmware.empty() && gpus_[gpu]->st.vbios.empty())
      incident(GM_HEALTH_WATCH_FIRMWARE, GM_HEALTH_WARN, "Firmware and VBIOS versions could not be read.");

    {
      std::lock_guard<std::mutex> l(state_mu_);
      health_base_[{g, gpu}] = next;
    }
    out->push_back(r);
  }
  return GM_ST_OK;
}

}  // namespace gpumon
