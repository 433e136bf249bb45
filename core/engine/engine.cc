// LocalEngine core: snapshot cache, groups, field groups, watches, sampler and
// event threads, discovery, status, introspection.  Health, policy and
// accounting live in their own translation units.
#include <sched.h>

#include "engine.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstring>
#include <unistd.h>

#include "../common/util.h"

namespace gpumon {

// ------------------------------------------------------------------ names --

const char* status_string(gm_status_t st) {
  switch (st) {
    case GM_ST_OK: return "Success";
    case GM_ST_BADPARAM: return "Bad parameter";
    case GM_ST_GENERIC_ERROR: return "Generic unspecified error";
    case GM_ST_MEMORY: return "Out of memory";
    case GM_ST_NOT_CONFIGURED: return "Setting not configured";
    case GM_ST_NOT_SUPPORTED: return "Feature not supported";
    case GM_ST_INIT_ERROR: return "Initialization error";
    case GM_ST_LIBRARY_NOT_FOUND: return "Device library (libamd_smi.so) not found";
    case GM_ST_FUNCTION_NOT_FOUND: return "Function not found in the device library";
    case GM_ST_NO_PERMISSION: return "Insufficient permission";
    case GM_ST_UNINITIALIZED: return "Engine not initialized";
    case GM_ST_NOT_WATCHED: return "Field is not being watched";
    case GM_ST_NO_DATA: return "No data is available";
    case GM_ST_STALE_DATA: return "Data is stale";
    case GM_ST_CONNECTION_NOT_VALID: return "Host engine connection is not valid";
    case GM_ST_GPU_NOT_SUPPORTED: return "This GPU is not supported";
    case GM_ST_GROUP_INCOMPATIBLE: return "GPUs are incompatible with the group";
    case GM_ST_MAX_LIMIT: return "Max limit reached for the object";
    case GM_ST_VER_MISMATCH: return "Structure version mismatch";
    case GM_ST_TIMEOUT: return "Timed out";
    case GM_ST_DUPLICATE_KEY: return "Duplicate key";
    case GM_ST_IN_USE: return "Object is in use";
    case GM_ST_NOT_FOUND: return "Object not found";
    case GM_ST_CHILD_SPAWN_FAILED: return "Could not start the host engine process";
    case GM_ST_INSUFFICIENT_SIZE: return "Buffer too small";
  }
  return "Unknown error";
}

const char* link_type_name(int32_t link, uint32_t hops) {
  static const char* xgmi[] = {"XGMI", "XGMI1", "XGMI2", "XGMI3", "XGMI4"};
  switch (link) {
    case GM_LINK_CROSS_CPU: return "SYS";
    case GM_LINK_SAME_CPU: return "NODE";
    case GM_LINK_HOST_BRIDGE: return "PHB";
    case GM_LINK_MULTI_SWITCH: return "PXB";
    case GM_LINK_SINGLE_SWITCH: return "PIX";
    case GM_LINK_SAME_BOARD: return "PSB";
    case GM_LINK_XGMI: return xgmi[hops <= 4 ? hops : 0];
  }
  return "N/A";
}

const char* link_type_description(int32_t link) {
  switch (link) {
    case GM_LINK_CROSS_CPU: return "Cross CPU socket";
    case GM_LINK_SAME_CPU: return "Same CPU socket";
    case GM_LINK_HOST_BRIDGE: return "Host PCI bridge";
    case GM_LINK_MULTI_SWITCH: return "Multiple PCI switches";
    case GM_LINK_SINGLE_SWITCH: return "Single PCI switch";
    case GM_LINK_SAME_BOARD: return "Same board";
    case GM_LINK_XGMI: return "xGMI (Infinity Fabric) link";
  }
  return "N/A";
}

const char* health_system_name(uint32_t s) {
  switch (s) {
    case GM_HEALTH_WATCH_PCIE: return "PCIe watches";
    case GM_HEALTH_WATCH_XGMI: return "xGMI watches";
    case GM_HEALTH_WATCH_MEM: return "Memory watches";
    case GM_HEALTH_WATCH_COMPUTE: return "Compute (XCD) watches";
    case GM_HEALTH_WATCH_THERMAL: return "Thermal watches";
    case GM_HEALTH_WATCH_POWER: return "Power watches";
    case GM_HEALTH_WATCH_DRIVER: return "Driver / RAS event watches";
    case GM_HEALTH_WATCH_FIRMWARE: return "Firmware watches";
  }
  return "N/A";
}

const char* health_result_name(int32_t h) {
  switch (h) {
    case GM_HEALTH_PASS: return "Healthy";
    case GM_HEALTH_WARN: return "Warning";
    case GM_HEALTH_FAIL: return "Failure";
  }
  return "N/A";
}

const char* policy_condition_name(uint32_t c) {
  switch (c) {
    case GM_POLICY_COND_ECC_UE: return "Uncorrectable ECC error";
    case GM_POLICY_COND_PCIE: return "PCI error";
    case GM_POLICY_COND_MAX_BAD_PAGES: return "Max Bad Pages Limit";
    case GM_POLICY_COND_THERMAL: return "Thermal Limit";
    case GM_POLICY_COND_POWER: return "Power Limit";
    case GM_POLICY_COND_XGMI: return "xGMI Error";
    case GM_POLICY_COND_GPU_EVENT: return "GPU Event (reset / VM fault / RAS)";
  }
  return "N/A";
}

const char* event_code_name(uint32_t code) {
  switch (code) {
    case GM_EVENT_NONE: return "none";
    case GM_EVENT_VMFAULT: return "VM fault";
    case GM_EVENT_THERMAL_THROTTLE: return "thermal throttle";
    case GM_EVENT_GPU_PRE_RESET: return "GPU reset started";
    case GM_EVENT_GPU_POST_RESET: return "GPU reset finished";
    case GM_EVENT_RAS_UE: return "RAS uncorrectable error";
    case GM_EVENT_RAS_POISON: return "RAS poison consumed";
    case GM_EVENT_RING_HANG: return "ring hang";
    case GM_EVENT_DIAG_FAIL: return "diagnostic failed";
    case GM_EVENT_PROCESS_START: return "process start";
    case GM_EVENT_PROCESS_END: return "process end";
  }
  return "unknown";
}

static bool is_critical_event(uint32_t code) {
  return code != GM_EVENT_NONE && code < GM_EVENT_PROCESS_START && code != GM_EVENT_THERMAL_THROTTLE;
}

void fill_attributes(const GpuStatic& s, gm_device_attributes_t* a) {
  uint32_t ver = a->version;
  memset(a, 0, sizeof *a);
  a->version = ver;
  a->gpu = s.index;
  a->supported = s.supported ? 1 : 0;
  copy_str(a->uuid, s.uuid);
  copy_str(a->name, s.name);
  copy_str(a->brand, s.brand);
  copy_str(a->serial, s.serial);
  copy_str(a->vbios, s.vbios);
  copy_str(a->firmware, s.firmware);
  copy_str(a->driver_version, s.driver_version);
  copy_str(a->pci_bdf, s.pci_bdf);
  copy_str(a->gfx_target, s.gfx_target);
  copy_str(a->device_path, s.device_path);
  copy_str(a->cpu_affinity, s.cpu_affinity);
  copy_str(a->compute_partition, s.compute_partition);
  copy_str(a->memory_partition, s.memory_partition);
  a->numa_node = s.numa_node;
  a->hip_id = s.hip_id;
  a->drm_render_minor = s.drm_render_minor;
  a->num_xcd = s.num_xcd;
  a->num_cu = s.num_cu;
  a->pcie_max_gen = s.pcie_max_gen;
  a->pcie_max_width = s.pcie_max_width;
  a->pcie_bandwidth_mbs = pcie_bandwidth_mbs(s.pcie_max_gen, s.pcie_max_width);
  a->power_cap_w = s.power_cap_w;
  a->power_cap_default_w = s.power_cap_default_w;
  a->vram_total_mib = s.vram_total_mib;
  a->vis_vram_total_mib = s.vis_vram_total_mib;
  a->gfx_clock_max_mhz = s.gfx_clock_max_mhz;
  a->mem_clock_max_mhz = s.mem_clock_max_mhz;
}

// --------------------------------------------------------------- notifier --

Notifier::Notifier() : th_([this] { run(); }) {}

Notifier::~Notifier() {
  {
    std::lock_guard<std::mutex> l(mu_);
    stop_ = true;
  }
  cv_.notify_all();
  if (th_.joinable()) th_.join();
}

void Notifier::post(std::function<void()> task) {
  {
    std::lock_guard<std::mutex> l(mu_);
    if (q_.size() >= 4096) {
      q_.pop_front();
      dropped_++;
    }
    q_.push_back(std::move(task));
  }
  cv_.notify_one();
}

void Notifier::drain() {
  std::unique_lock<std::mutex> l(mu_);
  idle_cv_.wait(l, [&] { return (q_.empty() && !busy_) || stop_; });
}

void Notifier::run() {
  std::unique_lock<std::mutex> l(mu_);
  while (true) {
    cv_.wait(l, [&] { return stop_ || !q_.empty(); });
    if (q_.empty() && stop_) break;
    auto task = std::move(q_.front());
    q_.pop_front();
    busy_ = true;
    l.unlock();
    task();
    l.lock();
    busy_ = false;
    if (q_.empty()) idle_cv_.notify_all();
  }
  idle_cv_.notify_all();
}

// ------------------------------------------------------------ LocalEngine --

LocalEngine::LocalEngine(std::unique_ptr<Backend> backend) : backend_(std::move(backend)) {
  thresholds_ = health_default_thresholds();
}

LocalEngine::~LocalEngine() { stop(); }

gm_status_t LocalEngine::start() {
  if (!backend_) return GM_ST_BADPARAM;
  gm_status_t st = backend_->init();
  if (st != GM_ST_OK) return st;
  uint32_t n = backend_->device_count();
  gpus_.clear();
  for (uint32_t i = 0; i < n && i < GM_MAX_GPUS; i++) {
    std::unique_ptr<GpuState> g(new GpuState());
    if (backend_->static_info(i, &g->st) != GM_ST_OK) {
      g->st.index = i;
      g->st.supported = false;
    }
    g->cells.reset(new Cell[field_count()]);
    g->watch.assign(field_count(), SlotWatch());
    g->history.resize(field_count());
    gpus_.push_back(std::move(g));
  }
  // "No event yet" is a value (0), like an XID field that never fired.
  int64_t wall = now_us();
  for (auto& g : gpus_) {
    SlotValue v;
    v.i64 = 0;
    v.status = GM_VS_OK;
    std::lock_guard<std::mutex> l(g->write_mu);
    publish(*g, field_slot(GM_FI_LAST_EVENT_CODE), v, wall);
  }
  start_mono_ = monotonic_us();
  running_ = true;
  sampler_ = std::thread([this] { sampler_loop(); });
  events_ = std::thread([this] { event_loop(); });
  GM_LOGI("engine started: backend=%s gpus=%u", backend_->name(), n);
  return GM_ST_OK;
}

void LocalEngine::stop() {
  if (!running_.exchange(false)) {
    if (sampler_.joinable()) sampler_.join();
    if (events_.joinable()) events_.join();
    return;
  }
  {
    std::lock_guard<std::mutex> l(wake_mu_);
    force_ = true;
  }
  wake_cv_.notify_all();
  cycle_cv_.notify_all();
  if (sampler_.joinable()) sampler_.join();
  backend_->shutdown();  // wakes a blocked poll_events
  if (events_.joinable()) events_.join();
  notifier_.drain();
}

// Writers hold g.write_mu.  One seq bump pair may cover many cells (see
// sample_gpu); this single-cell form is for events and injection.
void LocalEngine::publish(GpuState& g, int slot, const SlotValue& v, int64_t ts) {
  if (slot < 0) return;
  uint64_t s = g.seq.load(std::memory_order_relaxed);
  g.seq.store(s + 1, std::memory_order_relaxed);
  std::atomic_thread_fence(std::memory_order_release);
  g.cells[slot].bits.store((uint64_t)v.i64, std::memory_order_relaxed);
  g.cells[slot].ts.store(ts, std::memory_order_relaxed);
  g.cells[slot].status.store(v.status, std::memory_order_relaxed);
  g.seq.store(s + 2, std::memory_order_release);
}

// Back off inside a seqlock retry loop: a pause for the first few tries, then give the writer the core.
static inline void seq_relax(int tries) {
#if defined(__x86_64__) || defined(__i386__)
  if (tries < 64) { __builtin_ia32_pause(); return; }
#endif
  sched_yield();
}
constexpr int kSeqTries = 200000;  // a writer holds the lock for microseconds; this many yields is seconds

bool LocalEngine::read_cell(GpuState& g, int slot, SlotValue* v, int64_t* ts) {
  for (int tries = 0; tries < kSeqTries; tries++) {
    uint64_t s1 = g.seq.load(std::memory_order_acquire);
    if (s1 & 1) { seq_relax(tries); continue; }
    uint64_t bits = g.cells[slot].bits.load(std::memory_order_relaxed);
    int64_t t = g.cells[slot].ts.load(std::memory_order_relaxed);
    uint8_t st = g.cells[slot].status.load(std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_acquire);
    if (g.seq.load(std::memory_order_relaxed) != s1) { seq_relax(tries); continue; }
    v->i64 = (int64_t)bits;
    v->status = st;
    if (ts) *ts = t;
    return true;
  }
  return false;
}

bool LocalEngine::get_i(uint32_t gpu, uint16_t id, int64_t* out) {
  int slot = field_slot(id);
  if (gpu >= gpus_.size() || slot < 0) return false;
  SlotValue v;
  if (!read_cell(*gpus_[gpu], slot, &v, nullptr) || v.status != GM_VS_OK) return false;
  *out = field_at(slot).type == GM_FT_DOUBLE ? (int64_t)llround(v.dbl) : v.i64;
  return true;
}

bool LocalEngine::get_d(uint32_t gpu, uint16_t id, double* out) {
  int slot = field_slot(id);
  if (gpu >= gpus_.size() || slot < 0) return false;
  SlotValue v;
  if (!read_cell(*gpus_[gpu], slot, &v, nullptr) || v.status != GM_VS_OK) return false;
  *out = field_at(slot).type == GM_FT_DOUBLE ? v.dbl : (double)v.i64;
  return true;
}

// -------------------------------------------------------------- discovery --

gm_status_t LocalEngine::device_count(uint32_t* n) {
  *n = (uint32_t)gpus_.size();
  return GM_ST_OK;
}

gm_status_t LocalEngine::supported_devices(std::vector<uint32_t>* ids) {
  ids->clear();
  for (auto& g : gpus_)
    if (g->st.supported) ids->push_back(g->st.index);
  return GM_ST_OK;
}

gm_status_t LocalEngine::device_attributes(uint32_t gpu, gm_device_attributes_t* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  fill_attributes(gpus_[gpu]->st, out);
  // The enforced power cap can change at run time; prefer the sampled value.
  double cap;
  if (get_d(gpu, GM_FI_DEV_POWER_CAP, &cap)) out->power_cap_w = cap;
  return GM_ST_OK;
}

gm_status_t LocalEngine::device_topology(uint32_t gpu, gm_topology_t* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  out->gpu = gpu;
  copy_str(out->cpu_affinity, gpus_[gpu]->st.cpu_affinity);
  out->numa_node = gpus_[gpu]->st.numa_node;
  for (uint32_t p = 0; p < gpus_.size(); p++) {
    if (p == gpu) continue;
    PeerLink pl;
    gm_status_t st = backend_->peer_link(gpu, p, &pl);
    if (st == GM_ST_NOT_SUPPORTED) {  // same as the reference: not supported -> empty list
      out->num_peers = 0;
      return GM_ST_OK;
    }
    if (st != GM_ST_OK) continue;
    gm_peer_link_t& o = out->peers[out->num_peers++];
    o.gpu = p;
    o.link = pl.link;
    o.hops = pl.hops;
    o.num_links = pl.num_links;
    o.weight = pl.weight;
    o.min_bw_mbs = pl.min_bw_mbs;
    o.max_bw_mbs = pl.max_bw_mbs;
    o.p2p_accessible = pl.p2p_accessible ? 1 : 0;
    // The PEER's bus id (the reference fills in the GPU's own: bindings/go/dcgm/topology.go:121-131).
    copy_str(o.pci_bdf, gpus_[p]->st.pci_bdf);
  }
  return GM_ST_OK;
}

// ----------------------------------------------------------------- groups --

gm_status_t LocalEngine::resolve_group(gm_group_t g, std::vector<uint32_t>* out) {
  out->clear();
  if (g == GM_GROUP_ALL_GPUS) {
    for (auto& s : gpus_)
      if (s->st.supported) out->push_back(s->st.index);
    return GM_ST_OK;
  }
  auto it = groups_.find(g);
  if (it == groups_.end()) return GM_ST_NOT_FOUND;
  *out = it->second.gpus;
  return GM_ST_OK;
}

gm_status_t LocalEngine::group_create(const std::string& name, gm_group_t* out) {
  std::lock_guard<std::mutex> l(state_mu_);
  if (groups_.size() >= 4096) return GM_ST_MAX_LIMIT;
  for (auto& kv : groups_)
    if (kv.second.name == name && !name.empty()) return GM_ST_DUPLICATE_KEY;
  gm_group_t id = next_group_++;
  groups_[id].name = name;
  *out = id;
  return GM_ST_OK;
}

gm_status_t LocalEngine::group_add(gm_group_t g, uint32_t gpu) {
  std::lock_guard<std::mutex> l(state_mu_);
  auto it = groups_.find(g);
  if (it == groups_.end()) return GM_ST_NOT_FOUND;
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  if (!gpus_[gpu]->st.supported) return GM_ST_GPU_NOT_SUPPORTED;
  auto& v = it->second.gpus;
  if (std::find(v.begin(), v.end(), gpu) != v.end()) return GM_ST_OK;
  v.push_back(gpu);
  std::sort(v.begin(), v.end());
  rebuild_watch_table();
  return GM_ST_OK;
}

gm_status_t LocalEngine::group_remove(gm_group_t g, uint32_t gpu) {
  std::lock_guard<std::mutex> l(state_mu_);
  auto it = groups_.find(g);
  if (it == groups_.end()) return GM_ST_NOT_FOUND;
  auto& v = it->second.gpus;
  auto p = std::find(v.begin(), v.end(), gpu);
  if (p == v.end()) return GM_ST_NOT_FOUND;
  v.erase(p);
  rebuild_watch_table();
  return GM_ST_OK;
}

gm_status_t LocalEngine::group_devices(gm_group_t g, std::vector<uint32_t>* ids) {
  std::lock_guard<std::mutex> l(state_mu_);
  return resolve_group(g, ids);
}

gm_status_t LocalEngine::group_destroy(gm_group_t g) {
  std::lock_guard<std::mutex> l(state_mu_);
  if (g == GM_GROUP_ALL_GPUS) return GM_ST_BADPARAM;
  if (!groups_.erase(g)) return GM_ST_NOT_FOUND;
  // Watches of a destroyed group end with it; policy registrations on it stay
  // alive on the GPUs they were armed for (the reference destroys the group while
  // its policy is live and loses it: bindings/go/dcgm/policy.go:387).
  watches_.erase(std::remove_if(watches_.begin(), watches_.end(), [&](const Watch& w) { return w.group == g; }),
                 watches_.end());
  for (auto it = health_base_.begin(); it != health_base_.end();)
    it = it->first.first == g ? health_base_.erase(it) : std::next(it);
  rebuild_watch_table();
  return GM_ST_OK;
}

gm_status_t LocalEngine::field_group_create(const std::vector<uint16_t>& ids, const std::string& name,
                                            gm_field_group_t* out) {
  if (ids.empty() || ids.size() > GM_MAX_FIELDS_PER_GROUP) return GM_ST_BADPARAM;
  for (uint16_t id : ids)
    if (!field_def(id)) return GM_ST_BADPARAM;
  std::lock_guard<std::mutex> l(state_mu_);
  for (auto& kv : field_group_names_)
    if (kv.second == name && !name.empty()) return GM_ST_DUPLICATE_KEY;
  gm_field_group_t id = next_fg_++;
  field_groups_[id] = ids;
  field_group_names_[id] = name;
  *out = id;
  return GM_ST_OK;
}

gm_status_t LocalEngine::field_group_destroy(gm_field_group_t fg) {
  std::lock_guard<std::mutex> l(state_mu_);
  if (!field_groups_.erase(fg)) return GM_ST_NOT_FOUND;
  field_group_names_.erase(fg);
  watches_.erase(std::remove_if(watches_.begin(), watches_.end(), [&](const Watch& w) { return w.fg == fg; }),
                 watches_.end());
  rebuild_watch_table();
  return GM_ST_OK;
}

// ---------------------------------------------------------------- watches --

gm_status_t LocalEngine::watch_fields(gm_group_t g, gm_field_group_t fg, int64_t freq_us, double keep_age_s,
                                      int32_t keep_samples) {
  if (freq_us < 1000 || keep_age_s < 0 || keep_samples < 0) return GM_ST_BADPARAM;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    std::vector<uint32_t> gp;
    gm_status_t st = resolve_group(g, &gp);
    if (st != GM_ST_OK) return st;
    auto f = field_groups_.find(fg);
    if (f == field_groups_.end()) return GM_ST_NOT_FOUND;
    Watch* existing = nullptr;
    for (auto& w : watches_)
      if (w.group == g && w.fg == fg) existing = &w;
    Watch nw;
    Watch& w = existing ? *existing : nw;
    w.internal = false;
    w.group = g;
    w.fg = fg;
    w.fields = f->second;
    w.freq_us = freq_us;
    w.keep_age_us = (int64_t)(keep_age_s * 1e6);
    w.keep_samples = keep_samples;
    if (!existing) watches_.push_back(nw);
    rebuild_watch_table();
  }
  wake_cv_.notify_all();
  return GM_ST_OK;
}

gm_status_t LocalEngine::unwatch_fields(gm_group_t g, gm_field_group_t fg) {
  std::lock_guard<std::mutex> l(state_mu_);
  size_t before = watches_.size();
  watches_.erase(std::remove_if(watches_.begin(), watches_.end(),
                                [&](const Watch& w) { return w.group == g && w.fg == fg; }),
                 watches_.end());
  if (before == watches_.size()) return GM_ST_NOT_WATCHED;
  rebuild_watch_table();
  return GM_ST_OK;
}

void LocalEngine::ensure_internal_watch(const std::string& key, const std::vector<uint32_t>& gpus,
                                        const std::vector<uint16_t>& fields, int64_t freq_us) {
  auto it = internal_watches_.find(key);
  if (it != internal_watches_.end()) {
    Watch& old = it->second;
    // Union of GPUs and fields, fastest frequency.
    for (uint32_t g : gpus)
      if (std::find(old.gpus.begin(), old.gpus.end(), g) == old.gpus.end()) old.gpus.push_back(g);
    for (uint16_t f : fields)
      if (std::find(old.fields.begin(), old.fields.end(), f) == old.fields.end()) old.fields.push_back(f);
    old.freq_us = std::min(old.freq_us, freq_us);
  } else {
    Watch w;
    w.internal = true;
    w.gpus = gpus;
    w.fields = fields;
    w.freq_us = freq_us;
    // The engine's own consumers (status, health, policy, accounting, the RAS baseline) read the latest value and
    // keep what they compare with themselves: two samples, not five minutes of them.  A caller's watch on the same
    // field that asks for more still gets it (rebuild_watch_table: the larger retention wins).
    w.keep_age_us = (int64_t)(GM_DEFAULT_MAX_KEEP_AGE_S * 1e6);
    w.keep_samples = 2;
    internal_watches_[key] = w;
  }
  rebuild_watch_table();
}

void LocalEngine::rebuild_watch_table() {
  const int nslots = field_count();
  for (auto& g : gpus_) {
    std::vector<SlotWatch> nw(nslots);
    std::vector<const Watch*> all;
    for (auto& w : watches_) all.push_back(&w);
    for (auto& kv : internal_watches_) all.push_back(&kv.second);
    for (const Watch* wp : all) {
      const Watch& w = *wp;
      std::vector<uint32_t> gp;
      if (w.internal) gp = w.gpus; else if (resolve_group(w.group, &gp) != GM_ST_OK) continue;
      if (std::find(gp.begin(), gp.end(), g->st.index) == gp.end()) continue;
      for (uint16_t id : w.fields) {
        int s = field_slot(id);
        if (s < 0 || field_at(s).source == SRC_STATIC || field_at(s).source == SRC_EVENTS) continue;
        SlotWatch& sw = nw[s];
        bool first = sw.freq_us == 0;
        sw.freq_us = first ? w.freq_us : std::min(sw.freq_us, w.freq_us);
        sw.keep_age_us = std::max(sw.keep_age_us, w.keep_age_us);
        // 0 means unlimited; unlimited wins.
        if (first) sw.keep_samples = w.keep_samples;
        else if (sw.keep_samples != 0) sw.keep_samples = w.keep_samples == 0 ? 0 : std::max(sw.keep_samples, w.keep_samples);
      }
    }
    // Keep the running schedule, but a newly added faster watch must not wait out
    // the slower period that was already in flight.
    const int64_t now = monotonic_us();
    for (int s = 0; s < nslots; s++)
      if (nw[s].freq_us && g->watch[s].freq_us && g->watch[s].next_due_mono)
        nw[s].next_due_mono = std::min(g->watch[s].next_due_mono, now + nw[s].freq_us);
    g->watch.swap(nw);
  }
  watch_gen_++;
}

gm_status_t LocalEngine::update_all(bool wait) {
  if (!running_) return GM_ST_UNINITIALIZED;
  std::unique_lock<std::mutex> l(wake_mu_);
  force_ = true;
  uint64_t target = ++force_requested_;
  wake_cv_.notify_all();
  if (!wait) return GM_ST_OK;
  // The sampler records the highest force request it has fully served.
  bool ok = cv_wait_for(cycle_cv_, l, std::chrono::seconds(30), [&] { return cycles_ >= target || !running_; });
  return ok ? GM_ST_OK : GM_ST_TIMEOUT;
}

void LocalEngine::sample_gpu(uint32_t gpu, uint32_t sources, const std::vector<uint8_t>& due,
                             const std::vector<SlotWatch>& watch, int64_t now_mono, int64_t now_wall) {
  GpuState& g = *gpus_[gpu];
  static thread_local Sample sample;
  sample.clear();
  gm_status_t st = backend_->sample(gpu, sources, &sample);
  if (st != GM_ST_OK) {
    backend_errors_++;
    // A failed call blanks what it was asked for; the sampler itself never stops.
    sample.blank_untouched(sources, GM_VS_ERROR);
  }
  const int nslots = field_count();
  // RAS events for platforms that do not hand them out: a step of the uncorrectable / deferred totals between two
  // ECC samples is the error having happened (the reference waits for the driver's XID event instead,
  // bindings/go/nvml/bindings.go:68-146).
  std::vector<DeviceEvent> synthesized;
  if ((sources & SRC_ECC) && !backend_->delivers_ras_events()) {
    struct { uint16_t id; int64_t* prev; uint32_t code; const char* what; } steps[] = {
        {GM_FI_ECC_UE_VOLATILE, &g.prev_ue, GM_EVENT_RAS_UE, "uncorrectable"},
        {GM_FI_ECC_DEFERRED, &g.prev_deferred, GM_EVENT_RAS_POISON, "deferred"}};
    for (auto& st2 : steps) {
      const int slot = field_slot(st2.id);
      if (slot < 0 || !sample.touched[slot] || sample.v[slot].status != GM_VS_OK) continue;
      const int64_t cur = sample.v[slot].i64;
      if (*st2.prev >= 0 && cur > *st2.prev) {
        DeviceEvent e;
        e.gpu = gpu;
        e.code = st2.code;
        e.ts_us = now_wall;
        e.message = strprintf("RAS: %lld %s error(s) since the last sample (total %lld)", (long long)(cur - *st2.prev), st2.what,
                              (long long)cur);  // fits the 63 characters of a violation payload
        synthesized.push_back(e);
      }
      *st2.prev = cur;
    }
  }
  {
  std::lock_guard<std::mutex> hl(g.hist_mu);
  {
    std::lock_guard<std::mutex> wl(g.write_mu);
    uint64_t s = g.seq.load(std::memory_order_relaxed);
    g.seq.store(s + 1, std::memory_order_relaxed);
    std::atomic_thread_fence(std::memory_order_release);
    for (int slot = 0; slot < nslots; slot++) {
      if (!sample.touched[slot]) continue;
      auto h = g.holds.find(slot);
      if (h != g.holds.end()) {
        if (h->second.until_mono_us && h->second.until_mono_us > now_mono) {
          sample.v[slot] = h->second.v;  // injected value still held
        } else {
          g.holds.erase(h);
        }
      }
      if (sample.v[slot].status == GM_VS_ERROR) backend_errors_++;
      g.cells[slot].bits.store((uint64_t)sample.v[slot].i64, std::memory_order_relaxed);
      g.cells[slot].ts.store(now_wall, std::memory_order_relaxed);
      g.cells[slot].status.store(sample.v[slot].status, std::memory_order_relaxed);
    }
    g.seq.store(s + 2, std::memory_order_release);
  }
  for (int slot = 0; slot < nslots; slot++) {
    if (!sample.touched[slot] || !due[slot]) continue;
    const SlotWatch& sw = watch[slot];
    auto& h = g.history[slot];
    h.push_back(HistoryPoint{now_wall, sample.v[slot]});
    if (sw.keep_samples > 0)
      while ((int32_t)h.size() > sw.keep_samples) h.pop_front();
    if (sw.keep_age_us > 0)
      while (!h.empty() && h.front().ts_us < now_wall - sw.keep_age_us) h.pop_front();
    if (h.size() > 400000) h.pop_front();  // hard cap: 100 Hz x 300 s plus slack
  }
  }  // hist_mu
  for (auto& e : synthesized) handle_event(e);
}

void LocalEngine::sampler_loop() {
  const int nslots = field_count();
  std::vector<std::vector<uint8_t>> due(gpus_.size(), std::vector<uint8_t>(nslots, 0));
  std::vector<uint32_t> sources(gpus_.size(), 0);
  // The sampler's own copy of the retention settings: control-plane calls replace
  // GpuState::watch under state_mu_ while a sample cycle runs outside that lock.
  std::vector<std::vector<SlotWatch>> watch_snap(gpus_.size());
  uint64_t snap_gen = ~0ull;
  while (running_) {
    bool forced;
    uint64_t serving;
    {
      std::lock_guard<std::mutex> l(wake_mu_);
      forced = force_;
      force_ = false;
      serving = force_requested_;
    }
    int64_t now_mono = monotonic_us();
    int64_t now_wall = now_us();
    int64_t next_deadline = now_mono + 1000000;  // idle re-check once a second
    bool any = false, pid_due = false;
    {
      std::lock_guard<std::mutex> l(state_mu_);
      const bool resnap = watch_gen_.load() != snap_gen;
      snap_gen = watch_gen_.load();
      bool wake_is_due = false;  // this wake-up is for a deadline (not for a changed watch table or the idle re-check)
      for (size_t gi = 0; gi < gpus_.size() && !wake_is_due; gi++)
        for (const SlotWatch& sw : gpus_[gi]->watch)
          if (sw.freq_us && sw.next_due_mono && sw.next_due_mono <= now_mono) { wake_is_due = true; break; }
      for (size_t gi = 0; gi < gpus_.size(); gi++) {
        GpuState& g = *gpus_[gi];
        if (resnap) watch_snap[gi] = g.watch;
        sources[gi] = 0;
        std::fill(due[gi].begin(), due[gi].end(), 0);
        for (int s = 0; s < nslots; s++) {
          SlotWatch& sw = g.watch[s];
          if (!sw.freq_us) continue;
          // A slot that comes due within a quarter of its own period is taken with this cycle: watches registered a
          // few milliseconds apart then share one cycle (one read of the bulk metrics table) instead of waking the
          // sampler once each.  The nominal schedule is kept, so the period stays exact.
          if (forced || sw.next_due_mono <= now_mono + (wake_is_due ? sw.freq_us / 4 : 0)) {
            due[gi][s] = 1;
            sources[gi] |= field_at(s).source;
            // Drift-free schedule; after a stall, restart from now.
            sw.next_due_mono = sw.next_due_mono ? sw.next_due_mono + sw.freq_us : now_mono + sw.freq_us;
            if (sw.next_due_mono <= now_mono) sw.next_due_mono = now_mono + sw.freq_us;
          }
          next_deadline = std::min(next_deadline, sw.next_due_mono);
        }
        if (sources[gi]) any = true;
      }
      for (auto& kv : pid_watch_) {
        PidWatch& pw = kv.second;
        if (forced || pw.next_due_mono <= now_mono) {
          pid_due = true;
          pw.next_due_mono = now_mono + pw.freq_us;
        }
        next_deadline = std::min(next_deadline, pw.next_due_mono);
      }
    }
    if (any || pid_due) {
      int64_t t0 = monotonic_us();
      for (size_t gi = 0; gi < gpus_.size(); gi++)
        if (sources[gi]) sample_gpu((uint32_t)gi, sources[gi], due[gi], watch_snap[gi], now_mono, now_wall);
      if (pid_due) accounting_tick(now_mono, now_wall);
      evaluate_policies(now_wall);
      int64_t dt = monotonic_us() - t0;
      last_cycle_us_ = dt;
      total_cycle_us_ += dt;
      timed_cycles_++;
    }
    {
      std::unique_lock<std::mutex> l(wake_mu_);
      if (forced) cycles_ = std::max(cycles_, serving);
      cycle_cv_.notify_all();
      if (!running_) break;
      if (force_) continue;
      uint64_t gen = watch_gen_.load();
      int64_t wait_us = next_deadline - monotonic_us();
      if (wait_us > 0)
        cv_wait_for(wake_cv_, l, std::chrono::microseconds(wait_us),
                          [&] { return force_ || !running_ || watch_gen_.load() != gen; });
    }
  }
}

// ------------------------------------------------------------------ events --

void LocalEngine::event_loop() {
  std::vector<DeviceEvent> evs;
  while (running_) {
    evs.clear();
    gm_status_t st = backend_->poll_events(250, &evs);
    if (st == GM_ST_NOT_SUPPORTED || st == GM_ST_NO_PERMISSION) {
      GM_LOGI("backend has no event channel (%s); event-driven policies rely on counters", status_string(st));
      return;
    }
    if (st != GM_ST_OK) {
      usleep(250000);
      continue;
    }
    for (auto& e : evs) handle_event(e);
  }
}

void LocalEngine::handle_event(const DeviceEvent& e) {
  if (e.gpu >= gpus_.size()) return;
  GpuState& g = *gpus_[e.gpu];
  int64_t ts = e.ts_us ? e.ts_us : now_us();
  if (is_critical_event(e.code)) {
    SlotValue v;
    v.i64 = e.code;
    v.status = GM_VS_OK;
    {
      std::lock_guard<std::mutex> l(g.write_mu);
      publish(g, field_slot(GM_FI_LAST_EVENT_CODE), v, ts);
    }
    std::lock_guard<std::mutex> l(g.hist_mu);
    g.critical_event_ts.push_back(ts);
    if (g.critical_event_ts.size() > 1024) g.critical_event_ts.erase(g.critical_event_ts.begin());
    g.last_event_code = e.code;
  }
  if (e.code == GM_EVENT_GPU_POST_RESET) backend_->refresh_handles();

  gm_event_t ev;
  memset(&ev, 0, sizeof ev);
  ev.gpu = e.gpu;
  ev.code = e.code;
  ev.ts_us = ts;
  copy_str(ev.uuid, g.st.uuid);
  copy_str(ev.message, e.message.empty() ? std::string(event_code_name(e.code)) : e.message);

  std::vector<EventFn> subs;
  std::vector<std::shared_ptr<PolicyReg>> regs;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    for (auto& kv : event_subs_) subs.push_back(kv.second);
    if (is_critical_event(e.code))
      for (auto& kv : policy_regs_) {
        if (!(kv.second->conditions & GM_POLICY_COND_GPU_EVENT)) continue;
        std::vector<uint32_t> gp;
        if (kv.second->arm.count(e.gpu)) regs.push_back(kv.second);
      }
  }
  for (auto& fn : subs) notifier_.post([fn, ev] { fn(ev); });
  for (auto& r : regs) {
    gm_violation_t v;
    memset(&v, 0, sizeof v);
    v.condition = GM_POLICY_COND_GPU_EVENT;
    v.gpu = e.gpu;
    v.ts_us = ts;
    v.d.event.code = e.code;
    copy_str(v.d.event.message, std::string(ev.message));
    fire(r, v);
  }
}

gm_status_t LocalEngine::event_subscribe(EventFn fn, uint32_t* sub) {
  {
    std::lock_guard<std::mutex> l(state_mu_);
    *sub = next_sub_++;
    event_subs_[*sub] = std::move(fn);
    // RAS events are derived from steps of the error totals where the platform does not hand them out: somebody
    // listening for events is a reason to keep those totals sampled.
    if (!backend_->delivers_ras_events()) {
      std::vector<uint32_t> all;
      for (uint32_t i = 0; i < gpus_.size(); i++) all.push_back(i);
      ensure_internal_watch("ras", all, {GM_FI_ECC_UE_VOLATILE, GM_FI_ECC_DEFERRED}, GM_DEFAULT_UPDATE_FREQ_US);
    }
  }
  // The totals seen now are the baseline: only later errors are events.  Best effort: the subscription stands
  // whether or not this cycle could run (an engine that is not sampling yet takes its baseline with its first cycle),
  // so the caller is never told "failed" about a callback that stays registered.
  gm_status_t st = update_all(true);
  if (st != GM_ST_OK) GM_LOGI("event subscription %u: baseline sample not taken (%s)", *sub, status_string(st));
  return GM_ST_OK;
}

gm_status_t LocalEngine::event_unsubscribe(uint32_t sub) {
  {
    std::lock_guard<std::mutex> l(state_mu_);
    if (!event_subs_.erase(sub)) return GM_ST_NOT_FOUND;
    // nobody listens any more: the error totals need not be sampled on the listeners' account
    if (event_subs_.empty() && internal_watches_.erase("ras")) rebuild_watch_table();
  }
  notifier_.drain();  // no callback of this subscription runs after we return
  return GM_ST_OK;
}

// ------------------------------------------------------- links, diagnostics --

gm_status_t LocalEngine::xgmi_links(uint32_t gpu, std::vector<gm_xgmi_link_t>* out) {
  out->clear();
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  std::vector<XgmiLink> links;
  gm_status_t st = backend_->xgmi_links(gpu, &links);
  if (st != GM_ST_OK) return st;
  for (auto& l : links) {
    gm_xgmi_link_t o;
    memset(&o, 0, sizeof o);
    o.link = l.index;
    o.state = l.state;
    o.peer_gpu = -1;
    if (!l.peer_bdf.empty())
      for (size_t p = 0; p < gpus_.size(); p++)
        if (p != gpu && gpus_[p]->st.pci_bdf == l.peer_bdf) o.peer_gpu = (int32_t)p;
    copy_str(o.peer_bdf, l.peer_bdf);
    o.bit_rate_gbps = l.bit_rate_gbps;
    o.max_bandwidth_gbps = l.max_bandwidth_gbps;
    o.rx_kb = l.rx_kb;
    o.tx_kb = l.tx_kb;
    out->push_back(o);
  }
  return GM_ST_OK;
}

gm_status_t LocalEngine::diag_submit(const gm_diag_result_t& r) {
  if (r.version != GM_DIAG_RESULT_VERSION) return GM_ST_VER_MISMATCH;
  if (r.gpu >= gpus_.size() || !r.test[0] || !memchr(r.test, 0, sizeof r.test) || !memchr(r.message, 0, sizeof r.message))
    return GM_ST_BADPARAM;
  if (r.health != GM_HEALTH_PASS && r.health != GM_HEALTH_WARN && r.health != GM_HEALTH_FAIL) return GM_ST_BADPARAM;
  if (!r.system || (r.system & (r.system - 1)) || !(r.system & GM_HEALTH_WATCH_ALL)) return GM_ST_BADPARAM;  // exactly one system
  // Test names go into label values, JSON and fixed-size reply arrays: a plain identifier, and a bounded number of them.
  for (const char* c = r.test; *c; c++)
    if (!isalnum((unsigned char)*c) && *c != '_' && *c != '.' && *c != '-') return GM_ST_BADPARAM;
  gm_diag_result_t kept = r;
  if (!kept.ts_us) kept.ts_us = now_us();
  GpuState& g = *gpus_[r.gpu];
  bool news;
  {
    std::lock_guard<std::mutex> l(g.hist_mu);
    auto it = g.diag.find(kept.test);
    if (it == g.diag.end() && g.diag.size() >= GM_MAX_DIAG_TESTS) return GM_ST_INSUFFICIENT_SIZE;
    // The same verdict handed in again (a report file read twice) is not a second failure.
    news = it == g.diag.end() || it->second.ts_us != kept.ts_us || it->second.health != kept.health;
    g.diag[kept.test] = kept;
  }
  if (news && kept.health == GM_HEALTH_FAIL) {
    DeviceEvent e;
    e.gpu = r.gpu;
    e.code = GM_EVENT_DIAG_FAIL;
    e.ts_us = kept.ts_us;
    e.message = strprintf("diagnostic %s failed: %s", kept.test, kept.message);
    handle_event(e);
  }
  return GM_ST_OK;
}

gm_status_t LocalEngine::diag_results(uint32_t gpu, std::vector<gm_diag_result_t>* out) {
  out->clear();
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  GpuState& g = *gpus_[gpu];
  std::lock_guard<std::mutex> l(g.hist_mu);
  for (auto& kv : g.diag) out->push_back(kv.second);
  return GM_ST_OK;
}

// ------------------------------------------------------------------ values --

void static_value(const GpuStatic& s, uint16_t id, gm_value_t* o) {
  auto str = [&](const std::string& x) {
    o->type = GM_FT_STRING;
    if (x.empty()) { o->status = GM_VS_NOT_SUPPORTED; return; }
    copy_str(o->v.str, x);
    o->status = GM_VS_OK;
  };
  auto i64 = [&](int64_t x, bool ok) {
    o->type = GM_FT_INT64;
    o->v.i64 = x;
    o->status = ok ? GM_VS_OK : GM_VS_NOT_SUPPORTED;
  };
  auto dbl = [&](double x) {
    o->type = GM_FT_DOUBLE;
    o->v.dbl = x;
    o->status = std::isnan(x) ? GM_VS_NOT_SUPPORTED : GM_VS_OK;
  };
  switch (id) {
    case GM_FI_DRIVER_VERSION: str(s.driver_version); break;
    case GM_FI_DEV_NAME: str(s.name); break;
    case GM_FI_DEV_BRAND: str(s.brand); break;
    case GM_FI_DEV_SERIAL: str(s.serial); break;
    case GM_FI_DEV_UUID: str(s.uuid); break;
    case GM_FI_DEV_PCI_BDF: str(s.pci_bdf); break;
    case GM_FI_DEV_VBIOS_VERSION: str(s.vbios); break;
    case GM_FI_DEV_FIRMWARE_VERSION: str(s.firmware); break;
    case GM_FI_DEV_GFX_TARGET: str(s.gfx_target); break;
    case GM_FI_DEV_CPU_AFFINITY: str(s.cpu_affinity); break;
    case GM_FI_DEV_COMPUTE_PARTITION: str(s.compute_partition); break;
    case GM_FI_DEV_MEMORY_PARTITION: str(s.memory_partition); break;
    case GM_FI_DEV_DRM_RENDER_MINOR: i64(s.drm_render_minor, s.drm_render_minor >= 0); break;
    case GM_FI_DEV_NUMA_NODE: i64(s.numa_node, s.numa_node >= 0); break;
    case GM_FI_DEV_NUM_XCD: i64(s.num_xcd, s.num_xcd > 0); break;
    case GM_FI_DEV_NUM_CU: i64(s.num_cu, s.num_cu > 0); break;
    case GM_FI_DEV_HIP_ID: i64(s.hip_id, s.hip_id >= 0); break;
    case GM_FI_DEV_OAM_ID: i64(s.oam_id, s.oam_id >= 0); break;
    case GM_FI_DEV_PARTITION_ID: i64(s.partition_id, true); break;
    case GM_FI_DEV_PCIE_MAX_GEN: i64(s.pcie_max_gen, s.pcie_max_gen > 0); break;
    case GM_FI_DEV_PCIE_MAX_WIDTH: i64(s.pcie_max_width, s.pcie_max_width > 0); break;
    case GM_FI_DEV_POWER_CAP_DEFAULT: dbl(s.power_cap_default_w); break;
    case GM_FI_DEV_POWER_CAP_MAX: dbl(s.power_cap_max_w); break;
    case GM_FI_DEV_GFX_CLOCK_MAX: i64(s.gfx_clock_max_mhz, s.gfx_clock_max_mhz >= 0); break;
    case GM_FI_DEV_MEM_CLOCK_MAX: i64(s.mem_clock_max_mhz, s.mem_clock_max_mhz >= 0); break;
    case GM_FI_DEV_VIS_VRAM_TOTAL: i64(s.vis_vram_total_mib, s.vis_vram_total_mib >= 0); break;
    default: o->status = GM_VS_NOT_SUPPORTED;
  }
}

gm_status_t LocalEngine::latest_values(uint32_t gpu, const uint16_t* ids, int n, gm_value_t* out) {
  if (gpu >= gpus_.size() || n < 0) return GM_ST_BADPARAM;
  GpuState& g = *gpus_[gpu];
  // One seqlock transaction for the whole batch: the values of one call belong
  // to one published state.
  bool consistent = false;
  for (int tries = 0; tries < kSeqTries && !consistent; tries++) {
    uint64_t s1 = g.seq.load(std::memory_order_acquire);
    if (s1 & 1) { seq_relax(tries); continue; }
    for (int i = 0; i < n; i++) {
      gm_value_t& o = out[i];
      o.field_id = ids[i];
      o.gpu = gpu;
      o.ts_us = 0;
      o.v.i64 = 0;
      o.status = GM_VS_BLANK;
      const FieldDef* f = field_def(ids[i]);
      if (!f) { o.type = 0; o.status = GM_VS_NOT_SUPPORTED; continue; }
      o.type = f->type;
      if (f->source == SRC_STATIC) continue;
      int slot = field_slot(ids[i]);
      o.v.i64 = (int64_t)g.cells[slot].bits.load(std::memory_order_relaxed);
      o.ts_us = g.cells[slot].ts.load(std::memory_order_relaxed);
      o.status = g.cells[slot].status.load(std::memory_order_relaxed);
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    consistent = g.seq.load(std::memory_order_relaxed) == s1;
    if (!consistent) seq_relax(tries);
  }
  if (!consistent) {  // never hand out a batch that may mix two published states
    for (int i = 0; i < n; i++) {
      out[i].status = GM_VS_BLANK;
      out[i].v.i64 = 0;
    }
    return GM_ST_STALE_DATA;
  }
  int64_t wall = 0;
  for (int i = 0; i < n; i++) {
    const FieldDef* f = field_def(ids[i]);
    if (f && f->source == SRC_STATIC) {
      if (!wall) wall = now_us();
      static_value(g.st, ids[i], &out[i]);
      out[i].ts_us = wall;
    }
  }
  return GM_ST_OK;
}

gm_status_t LocalEngine::values_since(uint32_t gpu, uint16_t id, int64_t since_us, std::vector<gm_value_t>* out) {
  out->clear();
  int slot = field_slot(id);
  if (gpu >= gpus_.size() || slot < 0) return GM_ST_BADPARAM;
  GpuState& g = *gpus_[gpu];
  {
    std::lock_guard<std::mutex> sl(state_mu_);
    if (!g.watch[slot].freq_us) return GM_ST_NOT_WATCHED;
  }
  std::lock_guard<std::mutex> l(g.hist_mu);
  for (auto& p : g.history[slot]) {
    if (p.ts_us <= since_us) continue;
    gm_value_t v;
    memset(&v, 0, sizeof v);
    v.field_id = id;
    v.type = field_at(slot).type;
    v.status = p.v.status;
    v.gpu = gpu;
    v.ts_us = p.ts_us;
    v.v.i64 = p.v.i64;
    out->push_back(v);
  }
  return GM_ST_OK;
}

gm_status_t LocalEngine::inject_value(uint32_t gpu, const gm_value_t& v, int32_t hold_ms) {
  int slot = field_slot(v.field_id);
  if (gpu >= gpus_.size() || slot < 0) return GM_ST_BADPARAM;
  const FieldDef& f = field_at(slot);
  if (f.type == GM_FT_STRING || f.source == SRC_STATIC) return GM_ST_NOT_SUPPORTED;
  GpuState& g = *gpus_[gpu];
  SlotValue sv;
  sv.i64 = v.v.i64;
  sv.status = v.status;
  int64_t ts = v.ts_us ? v.ts_us : now_us();
  {
    std::lock_guard<std::mutex> hl(g.hist_mu);
    g.holds[slot] = Hold{sv, hold_ms > 0 ? monotonic_us() + (int64_t)hold_ms * 1000 : 0};
    g.history[slot].push_back(HistoryPoint{ts, sv});
    std::lock_guard<std::mutex> wl(g.write_mu);
    publish(g, slot, sv, ts);
  }
  // Policies look at the cache: evaluate now so an injected crossing fires even
  // when nothing else is due.
  evaluate_policies(ts);
  return GM_ST_OK;
}

gm_status_t LocalEngine::fake_ctl(const std::string& cmd) { return backend_->control(cmd); }

// ------------------------------------------------------------------ status --

const uint16_t kStatusFields[] = {
    GM_FI_POWER_USAGE, GM_FI_GPU_TEMP, GM_FI_GFX_UTIL, GM_FI_MEM_UTIL, GM_FI_ENC_UTIL, GM_FI_DEC_UTIL,
    GM_FI_GFX_CLOCK, GM_FI_MEM_CLOCK, GM_FI_DEV_VIS_VRAM_USED, GM_FI_PCIE_RX_KB, GM_FI_PCIE_TX_KB,
    GM_FI_PCIE_REPLAY_COUNTER, GM_FI_VRAM_USED, GM_FI_VRAM_FREE, GM_FI_ECC_CE_VOLATILE, GM_FI_ECC_UE_VOLATILE,
    GM_FI_PERF_LEVEL, GM_FI_FAN_SPEED, GM_FI_THROTTLE_STATUS, GM_FI_MEM_TEMP, GM_FI_TOTAL_ENERGY};
const int kNumStatusFields = (int)(sizeof kStatusFields / sizeof kStatusFields[0]);

void fill_status(uint32_t gpu, const gm_value_t* vals, gm_device_status_t* out) {
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  out->gpu = gpu;
  int64_t* ints[] = {nullptr, &out->temperature_c, &out->gfx_util, &out->mem_util, &out->enc_util, &out->dec_util,
                     &out->gfx_clock_mhz, &out->mem_clock_mhz, &out->vis_vram_used_mib, &out->pcie_rx_kb,
                     &out->pcie_tx_kb, &out->pcie_replays, &out->vram_used_mib, &out->vram_free_mib, &out->ecc_ce,
                     &out->ecc_ue, &out->perf_level, &out->fan_speed, &out->throttle_status, &out->mem_temp_c,
                     &out->energy_mj};
  for (int i = 0; i < kNumStatusFields; i++) {
    if (vals[i].status != GM_VS_OK) continue;
    out->valid |= 1ull << i;
    if (i == GM_DS_POWER) out->power_w = vals[i].v.dbl; else *ints[i] = vals[i].v.i64;
  }
}

gm_status_t LocalEngine::device_status(uint32_t gpu, gm_device_status_t* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  if (!gpus_[gpu]->st.supported) return GM_ST_GPU_NOT_SUPPORTED;
  const int n = kNumStatusFields;
  bool fresh = false;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    std::string key = strprintf("status:%u", gpu);
    if (!internal_watches_.count(key)) {
      ensure_internal_watch(key, {gpu}, std::vector<uint16_t>(kStatusFields, kStatusFields + n),
                            GM_DEFAULT_UPDATE_FREQ_US);
      fresh = true;
    }
  }
  if (fresh) {
    gm_status_t st = update_all(true);
    if (st != GM_ST_OK) return st;
  }
  std::vector<gm_value_t> vals(n);
  gm_status_t st = latest_values(gpu, kStatusFields, n, vals.data());
  if (st != GM_ST_OK) return st;
  fill_status(gpu, vals.data(), out);
  return GM_ST_OK;
}

gm_status_t LocalEngine::device_processes(uint32_t gpu, std::vector<gm_process_t>* out) {
  out->clear();
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  std::vector<ProcSample> ps;
  gm_status_t st = backend_->processes(gpu, &ps);
  if (st != GM_ST_OK) return st;
  for (auto& p : ps) {
    gm_process_t o;
    memset(&o, 0, sizeof o);
    o.pid = p.pid;
    o.type = 0;  // CDNA parts run compute queues only
    o.vram_bytes = p.vram_bytes;
    o.gtt_bytes = p.gtt_bytes;
    o.gfx_engine_ns = p.gfx_engine_ns;
    o.cu_occupancy = p.cu_occupancy;
    std::string name = p.name.empty() ? process_name(p.pid) : p.name;
    copy_str(o.name, name);
    out->push_back(o);
  }
  return GM_ST_OK;
}

gm_status_t LocalEngine::process_source(uint32_t gpu, std::string* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  std::vector<ProcSample> ps;
  gm_status_t st = backend_->processes(gpu, &ps);  // the source is known once the list was asked for
  *out = backend_->process_source(gpu);
  return st == GM_ST_OK || st == GM_ST_NOT_SUPPORTED || st == GM_ST_NO_PERMISSION ? GM_ST_OK : st;
}

gm_status_t LocalEngine::metrics_table_source(uint32_t gpu, std::string* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  *out = backend_->metrics_table_source(gpu);
  return GM_ST_OK;
}

gm_status_t LocalEngine::device_mode(uint32_t gpu, gm_device_mode_t* out) {
  if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  out->gpu = gpu;
  copy_str(out->compute_partition, gpus_[gpu]->st.compute_partition);
  copy_str(out->memory_partition, gpus_[gpu]->st.memory_partition);
  std::lock_guard<std::mutex> l(state_mu_);
  out->accounting_enabled = pid_watch_.count(gpu) ? 1 : 0;
  out->accounting_buffer_size = GM_MAX_PROCS;
  return GM_ST_OK;
}

// ----------------------------------------------------------- introspection --

static double self_cpu_seconds() {
  std::string s;
  if (!read_file("/proc/self/stat", &s)) return 0;
  size_t p = s.rfind(')');
  if (p == std::string::npos) return 0;
  std::vector<std::string> f = split(trim(s.substr(p + 1)), ' ');
  if (f.size() < 13) return 0;
  double ticks = atof(f[11].c_str()) + atof(f[12].c_str());  // utime + stime
  return ticks / (double)sysconf(_SC_CLK_TCK);
}

gm_status_t LocalEngine::introspect_toggle(bool enabled) {
  if (enabled && !introspect_on_) {
    intro_t0_mono_ = monotonic_us();
    intro_cpu0_ = self_cpu_seconds();
  }
  introspect_on_ = enabled;
  return GM_ST_OK;
}

gm_status_t LocalEngine::introspect(gm_introspect_t* out) {
  if (!introspect_on_) return GM_ST_NOT_CONFIGURED;
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  std::string s;
  if (read_file("/proc/self/statm", &s)) {
    auto f = split(trim(s), ' ');
    if (f.size() >= 2) out->memory_kb = atoll(f[1].c_str()) * (sysconf(_SC_PAGESIZE) / 1024);
  }
  int64_t now = monotonic_us();
  double wall = (double)(now - intro_t0_mono_) / 1e6;
  out->cpu_percent = wall > 0 ? 100.0 * (self_cpu_seconds() - intro_cpu0_) / wall : 0;
  out->uptime_us = now - start_mono_;
  out->sample_cycles = timed_cycles_;
  out->backend_errors = backend_errors_;
  out->last_cycle_ms = (double)last_cycle_us_ / 1000.0;
  out->avg_cycle_ms = timed_cycles_ ? (double)total_cycle_us_ / 1000.0 / (double)timed_cycles_ : 0;
  return GM_ST_OK;
}

}  // namespace gpumon
