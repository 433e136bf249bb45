// libgpumon_prof.so: device-wide MFMA pipe activity (and the shader clock held) through rocprofiler-sdk's
// device counting service.  See prof_counters.h for why this is a library of its own.
//
// One counting context per GPU agent, created inside the tool-initialisation callback
// (the only place rocprofiler-sdk lets a context be created), with a few counter
// configurations it rotates through: each holds GRBM_COUNT and ONE busy signal of the GRBM
// block (it has two slots) plus SQ_VALU_MFMA_BUSY_CYCLES (one of the eight SQ slots).  A
// sample reads the counters, stops the context, switches to the next configuration and
// starts again, so every window begins from zero and a reader needs no deltas.
//
// What is and is not device-wide (measured on MI355X, ROCm 7.2, with the load in another
// process): the GRBM counters and the MFMA pipe's busy cycles count all work on the device;
// the SQ counters that count waves or their instructions (SQ_WAVES, SQ_BUSY_CYCLES,
// SQ_BUSY_CU_CYCLES, SQ_WAVE_CYCLES, SQ_LEVEL_WAVES) stay at zero for waves of other
// processes, whose queues do not carry the per-wave counting enable, and so do the L2's
// memory-request counters (TCC_EA0_RDREQ / WRREQ) under another process's HBM stream.  They are
// therefore not collected, and CU activity / wave occupancy / memory traffic are not reported from here.
#include "prof_counters.h"

#include <hsa/hsa.h>
#include <rocprofiler-sdk/registration.h>
#include <rocprofiler-sdk/rocprofiler.h>

#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {

enum Slot { kCount = 0, kGrbm, kMfmaBusy, kExtra0, kExtra1, kExtra2, kSlots };
// The last three are for looking at other SQ counters with gpumon-prof-probe ($GPUMON_PROF_EXTRA=A,B,C, summed
// over their instances); nothing is derived from them.
std::string g_extra[3];
// Rotation of the second GRBM slot, one entry per counter set.
std::vector<std::string> g_grbm = {"GRBM_GUI_ACTIVE", "GRBM_SPI_BUSY", "GRBM_TA_BUSY"};
std::string slot_counter(int set, int s) {
  if (s == kCount) return "GRBM_COUNT";
  if (s == kGrbm) return g_grbm[(size_t)set];
  if (s == kMfmaBusy) return "SQ_VALU_MFMA_BUSY_CYCLES";
  return g_extra[s - kExtra0];
}

// One counter configuration: GRBM_COUNT, one rotating GRBM counter, the matrix-pipe counter, the extras.
struct Set {
  rocprofiler_counter_config_id_t cfg{};
  std::map<uint64_t, int> slot_of;  // counter id -> Slot
  bool have[kSlots] = {};
  size_t max_records = 0;
};

struct Agent {
  rocprofiler_agent_v0_t info;
  rocprofiler_context_id_t ctx{};
  std::vector<Set> sets;
  std::vector<std::string> set_names;  // the rotating GRBM counter of each set
  size_t active = 0;  // the set the running window counts
  bool running = false;
  std::chrono::steady_clock::time_point started;
};

std::mutex g_mu;
std::vector<Agent> g_agents;
std::string g_err;
bool g_tool_inited = false, g_open = false, g_hsa = false;
int g_init_rc = 0;

void set_err(const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
}

#define RP(call)                                                                   \
  do {                                                                             \
    rocprofiler_status_t rp_st = (call);                                           \
    if (rp_st != ROCPROFILER_STATUS_SUCCESS) {                                     \
      set_err("%s: %s", #call, rocprofiler_get_status_string(rp_st));              \
      return -1;                                                                   \
    }                                                                              \
  } while (0)

void set_config_cb(rocprofiler_context_id_t ctx, rocprofiler_agent_id_t, rocprofiler_device_counting_agent_cb_t set_config,
                   void* user) {
  auto* a = static_cast<Agent*>(user);
  if (a && a->active < a->sets.size() && a->sets[a->active].cfg.handle) set_config(ctx, a->sets[a->active].cfg);
}

int setup_agent(Agent& a) {
  std::vector<rocprofiler_counter_id_t> all;
  RP(rocprofiler_iterate_agent_supported_counters(
      a.info.id,
      [](rocprofiler_agent_id_t, rocprofiler_counter_id_t* c, size_t n, void* u) {
        auto* v = static_cast<std::vector<rocprofiler_counter_id_t>*>(u);
        v->insert(v->end(), c, c + n);
        return ROCPROFILER_STATUS_SUCCESS;
      },
      &all));
  struct Known {
    rocprofiler_counter_id_t id;
    std::string name;
    size_t instances;
  };
  std::vector<Known> known;
  for (auto& c : all) {
    rocprofiler_counter_info_v1_t ci;
    if (rocprofiler_query_counter_info(c, ROCPROFILER_COUNTER_INFO_VERSION_1, &ci) == ROCPROFILER_STATUS_SUCCESS && ci.name)
      known.push_back({c, ci.name, ci.dimensions_instances_count});
  }
  for (size_t k = 0; k < g_grbm.size(); k++) {
    Set set;
    std::vector<rocprofiler_counter_id_t> want;
    for (int s = 0; s < kSlots; s++) {
      const std::string name = slot_counter((int)k, s);
      if (name.empty()) continue;
      for (auto& c : known)
        if (!set.have[s] && c.name == name) {
          set.have[s] = true;
          set.slot_of[c.id.handle] = s;
          set.max_records += c.instances;
          want.push_back(c.id);
        }
    }
    if (!set.have[kCount]) {
      set_err("agent %s has no GRBM_COUNT counter", a.info.name ? a.info.name : "?");
      return -1;
    }
    if (!set.have[kGrbm]) continue;  // a name this device does not know: leave the set out
    RP(rocprofiler_create_counter_config(a.info.id, want.data(), want.size(), &set.cfg));
    a.sets.push_back(std::move(set));
    a.set_names.push_back(g_grbm[k]);
  }
  if (a.sets.empty()) {
    set_err("agent %s knows none of the GRBM busy counters asked for", a.info.name ? a.info.name : "?");
    return -1;
  }
  RP(rocprofiler_create_context(&a.ctx));
  RP(rocprofiler_configure_device_counting_service(a.ctx, rocprofiler_buffer_id_t{0}, a.info.id, set_config_cb, &a));
  return 0;
}

int tool_init(rocprofiler_client_finalize_t, void*) {
  g_tool_inited = true;
  if (const char* e = getenv("GPUMON_PROF_EXTRA")) {
    std::string all = e;
    for (int i = 0; i < 3 && !all.empty(); i++) {
      const size_t c = all.find(',');
      g_extra[i] = all.substr(0, c);
      all = c == std::string::npos ? std::string() : all.substr(c + 1);
    }
  }
  if (const char* e = getenv("GPUMON_PROF_GRBM"); e && *e) {
    g_grbm.clear();
    std::string all = e;
    while (!all.empty() && g_grbm.size() < 8) {
      const size_t c = all.find(',');
      if (c != 0) g_grbm.push_back(all.substr(0, c));
      all = c == std::string::npos ? std::string() : all.substr(c + 1);
    }
    if (g_grbm.empty()) g_grbm = {"GRBM_GUI_ACTIVE"};
  }
  std::vector<rocprofiler_agent_v0_t> gpus;
  rocprofiler_status_t st = rocprofiler_query_available_agents(
      ROCPROFILER_AGENT_INFO_VERSION_0,
      [](rocprofiler_agent_version_t, const void** arr, size_t n, void* u) {
        auto* v = static_cast<std::vector<rocprofiler_agent_v0_t>*>(u);
        for (size_t i = 0; i < n; i++) {
          const auto* ag = static_cast<const rocprofiler_agent_v0_t*>(arr[i]);
          if (ag->type == ROCPROFILER_AGENT_TYPE_GPU) v->push_back(*ag);
        }
        return ROCPROFILER_STATUS_SUCCESS;
      },
      sizeof(rocprofiler_agent_v0_t), &gpus);
  if (st != ROCPROFILER_STATUS_SUCCESS) {
    set_err("rocprofiler_query_available_agents: %s", rocprofiler_get_status_string(st));
    g_init_rc = -1;
    return 0;  // never fail the host process: the fields just stay blank
  }
  g_agents.resize(gpus.size());  // sized once: set_config_cb keeps pointers to the elements
  for (size_t i = 0; i < gpus.size(); i++) {
    g_agents[i].info = gpus[i];
    if (setup_agent(g_agents[i]) != 0) {
      g_init_rc = -1;
      g_agents.clear();
      return 0;
    }
  }
  return 0;
}

void tool_fini(void*) {}

rocprofiler_tool_configure_result_t* configure(uint32_t, const char*, uint32_t, rocprofiler_client_id_t* id) {
  id->name = "gpumon-prof";
  static rocprofiler_tool_configure_result_t cfg{sizeof(rocprofiler_tool_configure_result_t), &tool_init, &tool_fini, nullptr};
  return &cfg;
}

void copy_err(char* err, int errlen) {
  if (err && errlen > 0) snprintf(err, (size_t)errlen, "%s", g_err.c_str());
}

}  // namespace

extern "C" {

int gmp_open(char* err, int errlen) {
  std::lock_guard<std::mutex> l(g_mu);
  if (g_open) return (int)g_agents.size();
  rocprofiler_status_t st = rocprofiler_force_configure(&configure);
  if (st != ROCPROFILER_STATUS_SUCCESS || !g_tool_inited) {
    // Most likely the compute runtime was already running in this process: the counting service has to be
    // registered before HSA starts.
    set_err("rocprofiler_force_configure: %s", rocprofiler_get_status_string(st));
    copy_err(err, errlen);
    return -1;
  }
  if (g_init_rc != 0) {
    copy_err(err, errlen);
    return -1;
  }
  hsa_status_t hs = hsa_init();
  if (hs != HSA_STATUS_SUCCESS) {
    const char* msg = nullptr;
    hsa_status_string(hs, &msg);
    set_err("hsa_init: %s", msg ? msg : "failed");
    copy_err(err, errlen);
    return -1;
  }
  g_hsa = true;
  g_open = true;
  return (int)g_agents.size();
}

int gmp_agent_info(int agent, gmp_agent_info_t* out) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_open || agent < 0 || agent >= (int)g_agents.size() || !out) return -1;
  const auto& i = g_agents[agent].info;
  memset(out, 0, sizeof *out);
  out->pci_domain = i.domain;
  out->pci_location = i.location_id;
  out->cu_count = i.cu_count;
  out->simd_count = i.simd_count;
  snprintf(out->name, sizeof out->name, "%s", i.name ? i.name : "");
  return 0;
}

int gmp_sample(int agent, gmp_sample_t* out) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_open || agent < 0 || agent >= (int)g_agents.size() || !out) {
    set_err("gmp_sample: not open or bad agent %d", agent);
    return -1;
  }
  Agent& a = g_agents[agent];
  const auto now = std::chrono::steady_clock::now();
  if (!a.running) {
    RP(rocprofiler_start_context(a.ctx));
    a.running = true;
    a.started = now;
    return 1;
  }
  const Set& set = a.sets[a.active];
  const std::string grbm_name = a.set_names[a.active];
  std::vector<rocprofiler_counter_record_t> rec(set.max_records + 16);
  size_t n = rec.size();
  rocprofiler_status_t st =
      rocprofiler_sample_device_counting_service(a.ctx, rocprofiler_user_data_t{}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &n);
  const double window_s = std::chrono::duration<double>(now - a.started).count();
  // Stop, move on to the next counter set, start: every window begins from zero, and one bad read does not end
  // the series.  (The configuration callback runs inside the start and picks sets[active].)
  rocprofiler_stop_context(a.ctx);
  a.running = false;
  a.active = (a.active + 1) % a.sets.size();
  if (rocprofiler_start_context(a.ctx) == ROCPROFILER_STATUS_SUCCESS) {
    a.running = true;
    a.started = std::chrono::steady_clock::now();
  }
  if (st != ROCPROFILER_STATUS_SUCCESS) {
    set_err("rocprofiler_sample_device_counting_service: %s", rocprofiler_get_status_string(st));
    return -1;
  }
  // The GRBM counters have one instance per XCC: GRBM_COUNT ticks once a clock in each (take the largest), the
  // busy counter is averaged over its instances, so one busy XCC of eight reads 12.5 %, not 100 %.  The SQ
  // counters add up over the chip.
  double v[kSlots] = {};
  int grbm_instances = 0;
  for (size_t i = 0; i < n; i++) {
    rocprofiler_counter_id_t cid{};
    if (rocprofiler_query_record_counter_id(rec[i].id, &cid) != ROCPROFILER_STATUS_SUCCESS) continue;
    auto it = set.slot_of.find(cid.handle);
    if (it == set.slot_of.end()) continue;
    const int s = it->second;
    if (s == kCount) v[s] = std::max(v[s], rec[i].counter_value);
    else v[s] += rec[i].counter_value;
    if (s == kGrbm) grbm_instances++;
  }
  const double nan = std::nan("");
  memset(out, 0, sizeof *out);
  for (int s = 0; s < kSlots; s++) out->raw[s] = v[s];
  out->window_s = window_s;
  out->gpu_active = out->mfma_active = out->clock_mhz = out->cu_active = out->mem_pipe_active = out->grbm_busy = nan;
  snprintf(out->grbm_name, sizeof out->grbm_name, "%s", grbm_name.c_str());
  out->grbm_instances = grbm_instances;
  const double cycles = v[kCount];
  if (cycles <= 0) return 0;  // a window without a clock tick: nothing to divide by
  auto pct = [](double x) { return x < 0 ? 0.0 : x > 100 ? 100.0 : x; };
  if (grbm_instances > 0) {
    out->grbm_busy = pct(100.0 * v[kGrbm] / (grbm_instances * cycles));
    if (grbm_name == "GRBM_GUI_ACTIVE") out->gpu_active = out->grbm_busy;
    else if (grbm_name == "GRBM_SPI_BUSY") out->cu_active = out->grbm_busy;
    else if (grbm_name == "GRBM_TA_BUSY") out->mem_pipe_active = out->grbm_busy;
  }
  if (window_s > 0) out->clock_mhz = cycles / window_s / 1e6;
  // SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs (the normalisation of rocprofiler-sdk's MfmaUtil,
  // taken over all cycles of the window instead of the busy ones).
  const double simds = a.info.simd_count;
  if (set.have[kMfmaBusy] && simds > 0) out->mfma_active = pct(100.0 * v[kMfmaBusy] / (simds * cycles));
  return 0;
}

int gmp_abi_version(void) { return GMP_ABI_VERSION; }

int gmp_set_count(void) {
  std::lock_guard<std::mutex> l(g_mu);
  return g_open && !g_agents.empty() ? (int)g_agents[0].sets.size() : 0;
}

const char* gmp_last_error(void) { return g_err.c_str(); }

void gmp_close(void) {
  std::lock_guard<std::mutex> l(g_mu);
  if (!g_open) return;
  for (auto& a : g_agents)
    if (a.running) {
      rocprofiler_stop_context(a.ctx);
      a.running = false;
    }
  // HSA and the registration stay for the life of the process: rocprofiler-sdk cannot be configured twice.
}

}  // extern "C"
