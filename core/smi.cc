// Direct device layer (gm_smi_*): the device library behind a thin, stateless
// API — no host engine, no sampler thread, no cache.  Every read is one
// synchronous round through the backend.  This is the counterpart of the
// reference's NVML package (bindings/go/nvml/nvml.go:274-604: Init/Shutdown,
// NewDevice, Status, GetP2PLink, GetDeviceMode, event sets in bindings.go:68-146),
// kept separate from the engine API the way that package is separate from dcgm.
#include <cstring>
#include <deque>
#include <mutex>

#include "backend/backend.h"
#include "common/util.h"
#include "engine/engine.h"
#include "gpumon.h"

using namespace gpumon;

namespace {

struct Smi {
  std::mutex mu;       // lifetime + static cache
  std::mutex call_mu;  // one device-library round at a time (Sample scratch is shared)
  int refs = 0;
  std::unique_ptr<Backend> be;
  std::vector<GpuStatic> st;
  Sample scratch;
  std::mutex ev_mu;
  std::deque<DeviceEvent> pending;
  bool events_unsupported = false;
};

Smi& S() {
  static Smi* s = new Smi();  // outlives static destructors of client code
  return *s;
}

#define SMI_READY(s)                                   \
  std::lock_guard<std::mutex> _l((s).mu);              \
  if (!(s).refs) return GM_ST_UNINITIALIZED

}  // namespace

extern "C" {

gm_status_t gm_smi_init(const char* backend, const char* backend_spec) {
  Smi& s = S();
  std::lock_guard<std::mutex> l(s.mu);
  if (s.refs > 0) {
    s.refs++;
    return GM_ST_OK;
  }
  const char* env_backend = getenv("GPUMON_BACKEND");
  const char* env_spec = getenv("GPUMON_BACKEND_SPEC");
  std::string name = backend ? backend : (env_backend ? env_backend : "amdsmi");
  std::string spec = backend_spec ? backend_spec : (env_spec ? env_spec : "");
  std::unique_ptr<Backend> b = make_backend(name, spec);
  if (!b) return GM_ST_BADPARAM;
  gm_status_t st = b->init();
  if (st != GM_ST_OK) return st;
  s.st.assign(b->device_count(), GpuStatic());
  for (uint32_t i = 0; i < s.st.size(); i++) {
    b->static_info(i, &s.st[i]);
    s.st[i].index = i;
  }
  s.be = std::move(b);
  s.events_unsupported = false;
  s.pending.clear();
  s.refs = 1;
  return GM_ST_OK;
}

gm_status_t gm_smi_shutdown(void) {
  Smi& s = S();
  std::lock_guard<std::mutex> l(s.mu);
  if (!s.refs) return GM_ST_UNINITIALIZED;
  if (--s.refs == 0) {
    std::lock_guard<std::mutex> c(s.call_mu);
    s.be->shutdown();
    s.be.reset();
    s.st.clear();
  }
  return GM_ST_OK;
}

gm_status_t gm_smi_device_count(uint32_t* count) {
  Smi& s = S();
  SMI_READY(s);
  if (!count) return GM_ST_BADPARAM;
  *count = (uint32_t)s.st.size();
  return GM_ST_OK;
}

gm_status_t gm_smi_driver_version(char* buf, size_t cap) {
  Smi& s = S();
  SMI_READY(s);
  if (!buf || !cap) return GM_ST_BADPARAM;
  if (s.st.empty() || s.st[0].driver_version.empty()) return GM_ST_NOT_SUPPORTED;
  copy_str(buf, cap, s.st[0].driver_version);
  return GM_ST_OK;
}

gm_status_t gm_smi_runtime_version(uint32_t* major, uint32_t* minor, char* buf, size_t cap) {
  Smi& s = S();
  SMI_READY(s);
  if (!major || !minor) return GM_ST_BADPARAM;
  std::string text;
  std::lock_guard<std::mutex> l(s.call_mu);
  gm_status_t st = s.be->runtime_version(major, minor, &text);
  if (st == GM_ST_OK && buf && cap) copy_str(buf, cap, text);
  return st;
}

gm_status_t gm_smi_device_attributes(uint32_t gpu, gm_device_attributes_t* out) {
  Smi& s = S();
  SMI_READY(s);
  if (!out || gpu >= s.st.size()) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_ATTRIBUTES_VERSION) return GM_ST_VER_MISMATCH;
  fill_attributes(s.st[gpu], out);
  return GM_ST_OK;
}

gm_status_t gm_smi_read_fields(uint32_t gpu, const uint16_t* ids, int n, gm_value_t* out) {
  Smi& s = S();
  SMI_READY(s);
  if (!ids || !out || n < 0 || gpu >= s.st.size()) return GM_ST_BADPARAM;
  uint32_t sources = 0;
  for (int i = 0; i < n; i++) {
    const FieldDef* f = field_def(ids[i]);
    if (f && f->source != SRC_STATIC && f->source != SRC_EVENTS) sources |= f->source;
  }
  std::lock_guard<std::mutex> c(s.call_mu);
  s.scratch.clear();
  gm_status_t st = sources ? s.be->sample(gpu, sources, &s.scratch) : GM_ST_OK;
  if (st != GM_ST_OK) s.scratch.blank_untouched(sources, GM_VS_ERROR);
  const int64_t now = now_us();
  for (int i = 0; i < n; i++) {
    gm_value_t& o = out[i];
    memset(&o, 0, sizeof o);
    o.field_id = ids[i];
    o.gpu = gpu;
    o.ts_us = now;
    o.status = GM_VS_NOT_SUPPORTED;
    const FieldDef* f = field_def(ids[i]);
    if (!f) continue;
    o.type = f->type;
    if (f->source == SRC_STATIC) {
      static_value(s.st[gpu], ids[i], &o);
      continue;
    }
    int slot = field_slot(ids[i]);
    if (slot < 0 || !s.scratch.touched[slot]) continue;
    o.v.i64 = s.scratch.v[slot].i64;
    o.status = s.scratch.v[slot].status;
  }
  return GM_ST_OK;
}

gm_status_t gm_smi_device_status(uint32_t gpu, gm_device_status_t* out) {
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_STATUS_VERSION) return GM_ST_VER_MISMATCH;
  {
    Smi& s = S();
    SMI_READY(s);
    if (gpu >= s.st.size()) return GM_ST_BADPARAM;
    if (!s.st[gpu].supported) return GM_ST_GPU_NOT_SUPPORTED;
  }
  std::vector<gm_value_t> vals(kNumStatusFields);
  gm_status_t st = gm_smi_read_fields(gpu, kStatusFields, kNumStatusFields, vals.data());
  if (st != GM_ST_OK) return st;
  fill_status(gpu, vals.data(), out);
  return GM_ST_OK;
}

gm_status_t gm_smi_device_processes(uint32_t gpu, gm_process_t* out, int* n) {
  Smi& s = S();
  SMI_READY(s);
  if (!out || !n || *n < 0 || gpu >= s.st.size()) return GM_ST_BADPARAM;
  std::vector<ProcSample> ps;
  gm_status_t st;
  {
    std::lock_guard<std::mutex> c(s.call_mu);
    st = s.be->processes(gpu, &ps);
  }
  if (st != GM_ST_OK) return st;
  const int cap = *n;
  *n = (int)ps.size();
  for (int i = 0; i < (int)ps.size() && i < cap; i++) {
    gm_process_t& o = out[i];
    memset(&o, 0, sizeof o);
    o.pid = ps[i].pid;
    o.vram_bytes = ps[i].vram_bytes;
    o.gtt_bytes = ps[i].gtt_bytes;
    o.gfx_engine_ns = ps[i].gfx_engine_ns;
    o.cu_occupancy = ps[i].cu_occupancy;
    // The process can exit between the list and the name lookup; an empty name is fine.
    copy_str(o.name, ps[i].name.empty() ? process_name(ps[i].pid) : ps[i].name);
  }
  return (int)ps.size() > cap ? GM_ST_INSUFFICIENT_SIZE : GM_ST_OK;
}

gm_status_t gm_smi_xgmi_links(uint32_t gpu, gm_xgmi_link_t* out, int* n) {
  Smi& s = S();
  SMI_READY(s);
  if (!n || gpu >= s.st.size()) return GM_ST_BADPARAM;
  std::vector<XgmiLink> links;
  gm_status_t st;
  {
    std::lock_guard<std::mutex> c(s.call_mu);
    st = s.be->xgmi_links(gpu, &links);
  }
  if (st != GM_ST_OK) return st;
  const int cap = *n;
  *n = (int)links.size();
  if (cap < (int)links.size() || (!out && !links.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < links.size(); i++) {
    gm_xgmi_link_t& o = out[i];
    memset(&o, 0, sizeof o);
    o.link = links[i].index;
    o.state = links[i].state;
    o.peer_gpu = -1;
    for (size_t p = 0; p < s.st.size(); p++)
      if (p != gpu && !links[i].peer_bdf.empty() && s.st[p].pci_bdf == links[i].peer_bdf) o.peer_gpu = (int32_t)p;
    copy_str(o.peer_bdf, links[i].peer_bdf);
    o.bit_rate_gbps = links[i].bit_rate_gbps;
    o.max_bandwidth_gbps = links[i].max_bandwidth_gbps;
    o.rx_kb = links[i].rx_kb;
    o.tx_kb = links[i].tx_kb;
  }
  return GM_ST_OK;
}

gm_status_t gm_smi_peer_link(uint32_t gpu, uint32_t peer, gm_peer_link_t* out) {
  Smi& s = S();
  SMI_READY(s);
  if (!out || gpu >= s.st.size() || peer >= s.st.size() || gpu == peer) return GM_ST_BADPARAM;
  PeerLink pl;
  gm_status_t st;
  {
    std::lock_guard<std::mutex> c(s.call_mu);
    st = s.be->peer_link(gpu, peer, &pl);
  }
  if (st != GM_ST_OK) return st;
  memset(out, 0, sizeof *out);
  out->gpu = peer;
  out->link = pl.link;
  out->hops = pl.hops;
  out->num_links = pl.num_links;
  out->weight = pl.weight;
  out->min_bw_mbs = pl.min_bw_mbs;
  out->max_bw_mbs = pl.max_bw_mbs;
  out->p2p_accessible = pl.p2p_accessible ? 1 : 0;
  copy_str(out->pci_bdf, s.st[peer].pci_bdf);
  return GM_ST_OK;
}

gm_status_t gm_smi_device_mode(uint32_t gpu, gm_device_mode_t* out) {
  Smi& s = S();
  SMI_READY(s);
  if (!out || gpu >= s.st.size()) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_MODE_VERSION) return GM_ST_VER_MISMATCH;
  uint32_t ver = out->version;
  memset(out, 0, sizeof *out);
  out->version = ver;
  out->gpu = gpu;
  out->accounting_enabled = 0;  // accounting lives in the host engine
  out->accounting_buffer_size = GM_MAX_PROCS;
  copy_str(out->compute_partition, s.st[gpu].compute_partition);
  copy_str(out->memory_partition, s.st[gpu].memory_partition);
  return GM_ST_OK;
}

gm_status_t gm_smi_event_wait(int32_t timeout_ms, gm_event_t* out) {
  Smi& s = S();
  if (!out) return GM_ST_BADPARAM;
  Backend* be;
  {
    std::lock_guard<std::mutex> l(s.mu);
    if (!s.refs) return GM_ST_UNINITIALIZED;
    be = s.be.get();
  }
  std::lock_guard<std::mutex> el(s.ev_mu);
  if (s.events_unsupported) return GM_ST_NOT_SUPPORTED;
  const int64_t deadline = monotonic_us() + (int64_t)(timeout_ms < 0 ? 0 : timeout_ms) * 1000;
  while (s.pending.empty()) {
    int64_t left_ms = (deadline - monotonic_us()) / 1000;
    if (left_ms <= 0 && timeout_ms > 0) return GM_ST_TIMEOUT;
    std::vector<DeviceEvent> evs;
    gm_status_t st = be->poll_events((int)std::max<int64_t>(0, std::min<int64_t>(left_ms, 250)), &evs);
    if (st == GM_ST_NOT_SUPPORTED || st == GM_ST_NO_PERMISSION) {
      s.events_unsupported = true;
      return GM_ST_NOT_SUPPORTED;
    }
    if (st != GM_ST_OK) return st;
    for (auto& e : evs) s.pending.push_back(e);
    if (s.pending.empty() && timeout_ms <= 0) return GM_ST_TIMEOUT;
  }
  DeviceEvent e = s.pending.front();
  s.pending.pop_front();
  memset(out, 0, sizeof *out);
  out->gpu = e.gpu;
  out->code = e.code;
  out->ts_us = e.ts_us ? e.ts_us : now_us();
  {
    std::lock_guard<std::mutex> l(s.mu);
    if (e.gpu < s.st.size()) copy_str(out->uuid, s.st[e.gpu].uuid);
  }
  copy_str(out->message, e.message);
  return GM_ST_OK;
}

/* Test hook of the direct layer: scripts the fake backend it owns. */
gm_status_t gm_smi_fake_ctl(const char* command) {
  Smi& s = S();
  SMI_READY(s);
  if (!command) return GM_ST_BADPARAM;
  std::lock_guard<std::mutex> c(s.call_mu);
  return s.be->control(command);
}

}  // extern "C"
