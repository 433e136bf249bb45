// libgpumon_prof.so: device-wide MFMA pipe activity (and the shader clock held) through rocprofiler-sdk's
// device counting service.  See prof_counters.h for why this is a library of its own.
//
// One counting context per GPU agent, created inside the tool-initialisation callback
// (the only place rocprofiler-sdk lets a context be created), each with one counter
// configuration of three hardware counters: GRBM_COUNT and GRBM_GUI_ACTIVE (both of the
// two GRBM slots) and SQ_VALU_MFMA_BUSY_CYCLES (one of the eight SQ slots).  A sample reads
// the counters, stops the context and starts it again, so every window begins from zero
// and a reader needs no deltas.
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

enum Slot { kCount = 0, kGuiActive, kMfmaBusy, kExtra0, kExtra1, kExtra2, kSlots };
// The last three are for looking at other SQ counters with gpumon-prof-probe ($GPUMON_PROF_EXTRA=A,B,C, summed
// over their instances); nothing is derived from them.
std::string g_extra[3];
const char* counter_name(int s) {
  static const char* const names[] = {"GRBM_COUNT", "GRBM_GUI_ACTIVE", "SQ_VALU_MFMA_BUSY_CYCLES"};
  return s < kExtra0 ? names[s] : g_extra[s - kExtra0].c_str();
}
// How the instances of a counter (one per XCC / shader engine / ...) fold into one number: the two GRBM
// counters tick once per clock in every XCC (take the largest), the SQ counters add up over the chip.
const bool kSum[kSlots] = {false, false, true, true, true, true};

struct Agent {
  rocprofiler_agent_v0_t info;
  rocprofiler_context_id_t ctx{};
  rocprofiler_counter_config_id_t cfg{};
  std::map<uint64_t, int> slot_of;  // counter id -> Slot
  bool have[kSlots] = {};
  size_t max_records = 0;
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
  if (a && a->cfg.handle) set_config(ctx, a->cfg);
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
  std::vector<rocprofiler_counter_id_t> want;
  for (auto& c : all) {
    rocprofiler_counter_info_v1_t ci;
    if (rocprofiler_query_counter_info(c, ROCPROFILER_COUNTER_INFO_VERSION_1, &ci) != ROCPROFILER_STATUS_SUCCESS || !ci.name)
      continue;
    for (int s = 0; s < kSlots; s++)
      if (!a.have[s] && !strcmp(ci.name, counter_name(s))) {
        a.have[s] = true;
        a.slot_of[c.handle] = s;
        a.max_records += ci.dimensions_instances_count;
        want.push_back(c);
      }
  }
  if (!a.have[kCount] || !a.have[kGuiActive]) {
    set_err("agent %s has no GRBM_COUNT / GRBM_GUI_ACTIVE counter", a.info.name ? a.info.name : "?");
    return -1;
  }
  RP(rocprofiler_create_counter_config(a.info.id, want.data(), want.size(), &a.cfg));
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
  std::vector<rocprofiler_counter_record_t> rec(a.max_records + 16);
  size_t n = rec.size();
  rocprofiler_status_t st =
      rocprofiler_sample_device_counting_service(a.ctx, rocprofiler_user_data_t{}, ROCPROFILER_COUNTER_FLAG_NONE, rec.data(), &n);
  const double window_s = std::chrono::duration<double>(now - a.started).count();
  static const bool keep_running = getenv("GPUMON_PROF_KEEP_RUNNING") != nullptr;  // experiment: no stop/start between windows
  if (!keep_running || st != ROCPROFILER_STATUS_SUCCESS) {
    // Restart whatever the read said, so that one bad read does not end the series.
    rocprofiler_stop_context(a.ctx);
    a.running = false;
    if (rocprofiler_start_context(a.ctx) == ROCPROFILER_STATUS_SUCCESS) {
      a.running = true;
      a.started = std::chrono::steady_clock::now();
    }
  } else {
    a.started = now;
  }
  if (st != ROCPROFILER_STATUS_SUCCESS) {
    set_err("rocprofiler_sample_device_counting_service: %s", rocprofiler_get_status_string(st));
    return -1;
  }
  double v[kSlots] = {};
  for (size_t i = 0; i < n; i++) {
    rocprofiler_counter_id_t cid{};
    if (rocprofiler_query_record_counter_id(rec[i].id, &cid) != ROCPROFILER_STATUS_SUCCESS) continue;
    auto it = a.slot_of.find(cid.handle);
    if (it == a.slot_of.end()) continue;
    const int s = it->second;
    v[s] = kSum[s] ? v[s] + rec[i].counter_value : std::max(v[s], rec[i].counter_value);
  }
  const double nan = std::nan("");
  memset(out->raw, 0, sizeof out->raw);
  for (int s = 0; s < kSlots; s++) out->raw[s] = v[s];
  out->window_s = window_s;
  out->gpu_active = out->mfma_active = out->clock_mhz = nan;
  const double cycles = v[kCount];
  if (cycles <= 0) return 0;  // a window without a clock tick: nothing to divide by
  auto pct = [](double x) { return x < 0 ? 0.0 : x > 100 ? 100.0 : x; };
  out->gpu_active = pct(100.0 * v[kGuiActive] / cycles);
  if (window_s > 0) out->clock_mhz = cycles / window_s / 1e6;
  // SQ_VALU_MFMA_BUSY_CYCLES is summed over the SIMDs (the normalisation of rocprofiler-sdk's MfmaUtil,
  // taken over all cycles of the window instead of the busy ones).
  const double simds = a.info.simd_count;
  if (a.have[kMfmaBusy] && simds > 0) out->mfma_active = pct(100.0 * v[kMfmaBusy] / (simds * cycles));
  return 0;
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
