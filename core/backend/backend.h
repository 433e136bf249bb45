// Device-layer interface.  Two implementations: AmdSmiBackend (libamd_smi.so
// through a dlopen'd function table) and FakeBackend (scripted telemetry with
// fault injection, used by every CPU test).
#pragma once

#include <cmath>
#include <cstdint>
#include <memory>
#include <string>
#include <vector>

#include "../fields.h"
#include "gpumon.h"

namespace gpumon {

// Static attributes of one GPU, read once at start-up.
struct GpuStatic {
  uint32_t index = 0;
  bool supported = true;
  std::string uuid, name, brand, serial, vbios, firmware, driver_version, pci_bdf, gfx_target;
  std::string device_path, cpu_affinity, compute_partition, memory_partition;
  int numa_node = -1;
  int hip_id = -1;
  int drm_render_minor = -1;
  int oam_id = -1;
  int partition_id = 0;  // index of this compute partition inside its GPU; 0 when not partitioned
  uint32_t num_xcd = 0, num_cu = 0;
  uint32_t pcie_max_gen = 0, pcie_max_width = 0;
  double power_cap_w = NAN, power_cap_default_w = NAN, power_cap_max_w = NAN;
  int64_t vram_total_mib = -1, vis_vram_total_mib = -1;
  int64_t gfx_clock_max_mhz = -1, mem_clock_max_mhz = -1;
};

// One sampled value; strings never travel through samples (static only).
struct SlotValue {
  union {
    int64_t i64;
    double dbl;
  };
  uint8_t status;  // gm_value_status_t
  SlotValue() : i64(0), status(GM_VS_BLANK) {}
};

// Output of Backend::sample: one SlotValue per registry slot.  A backend sets
// every field of each source it was asked for — to OK with a value, or to
// NOT_SUPPORTED / NO_PERMISSION / ERROR — and leaves the rest untouched.
struct Sample {
  std::vector<SlotValue> v;
  std::vector<uint8_t> touched;
  Sample() : v(field_count()), touched(field_count(), 0) {}
  void clear() { std::fill(touched.begin(), touched.end(), 0); }
  void set_i(uint16_t id, int64_t x) {
    int s = field_slot(id);
    if (s < 0) return;
    v[s].i64 = x;
    v[s].status = GM_VS_OK;
    touched[s] = 1;
  }
  void set_d(uint16_t id, double x) {
    int s = field_slot(id);
    if (s < 0) return;
    v[s].dbl = x;
    v[s].status = GM_VS_OK;
    touched[s] = 1;
  }
  // Stores an integer or a double according to the field's registered type.
  void set_num(uint16_t id, double x) {
    const FieldDef* f = field_def(id);
    if (!f) return;
    if (f->type == GM_FT_DOUBLE) set_d(id, x); else set_i(id, (int64_t)llround(x));
  }
  void set_blank(uint16_t id, uint8_t why) {
    int s = field_slot(id);
    if (s < 0) return;
    v[s].i64 = 0;
    v[s].status = why;
    touched[s] = 1;
  }
  // Marks every field of the given sources that was not set as blank(why).
  void blank_untouched(uint32_t sources, uint8_t why) {
    for (int s = 0; s < field_count(); s++)
      if (!touched[s] && (field_at(s).source & sources) && field_at(s).type != GM_FT_STRING) {
        v[s].status = why;
        v[s].i64 = 0;
        touched[s] = 1;
      }
  }
};

struct ProcSample {
  uint32_t pid = 0;
  std::string name;
  uint64_t vram_bytes = 0, gtt_bytes = 0, cpu_mem_bytes = 0;  // ~0 = not known (GM_PROC_MEM_UNKNOWN)
  uint64_t gfx_engine_ns = 0, enc_engine_ns = 0;
  uint32_t cu_occupancy = 0xffffffffu;
};

struct PeerLink {
  gm_link_type_t link = GM_LINK_UNKNOWN;
  uint32_t hops = 0;
  uint32_t num_links = 0;
  uint64_t weight = 0;
  int64_t min_bw_mbs = -1, max_bw_mbs = -1;
  bool p2p_accessible = false;
};

// One physical xGMI link (see gm_xgmi_link_t).
struct XgmiLink {
  uint32_t index = 0;
  int state = -1;        // 1 up, 0 down, -1 not wired / not reported
  std::string peer_bdf;  // empty when unknown
  int64_t bit_rate_gbps = -1, max_bandwidth_gbps = -1, rx_kb = -1, tx_kb = -1;
};

struct DeviceEvent {
  uint32_t gpu = 0;
  uint32_t code = GM_EVENT_NONE;
  int64_t ts_us = 0;
  std::string message;
};

// Per-device duration from per-XCC residency accumulators (~0 = not reported): the XCC that spent longest in the
// state.  A sum would let a "duration" of the device advance several seconds per wall-clock second.
inline bool device_residency(const uint64_t* per_xcc, int n, uint64_t* out) {
  bool any = false;
  uint64_t worst = 0;
  for (int i = 0; i < n; i++) {
    if (per_xcc[i] == ~0ull) continue;
    worst = any ? (per_xcc[i] > worst ? per_xcc[i] : worst) : per_xcc[i];
    any = true;
  }
  if (any) *out = worst;
  return any;
}

class Backend {
 public:
  virtual ~Backend() {}
  virtual const char* name() const = 0;
  virtual gm_status_t init() = 0;
  virtual void shutdown() = 0;
  virtual uint32_t device_count() = 0;
  virtual gm_status_t static_info(uint32_t gpu, GpuStatic* out) = 0;
  // Fills `out` for the requested Source bits; never throws, never blocks long.
  virtual gm_status_t sample(uint32_t gpu, uint32_t sources, Sample* out) = 0;
  virtual gm_status_t processes(uint32_t gpu, std::vector<ProcSample>* out) = 0;
  // Where the last processes() answer for this GPU came from: "kfd" (the device library's table), "fdinfo"
  // (the kernel's per-file DRM statistics), "open-files" (holders of the render node, no figures), "fake", "none".
  virtual const char* process_source(uint32_t gpu) { (void)gpu; return "none"; }
  // How the bulk metrics table of this GPU is read: "direct" (the backend's own read of the driver's file),
  // "library", with a reason after a colon where direct reads were refused; "n/a" for backends without one.
  virtual std::string metrics_table_source(uint32_t gpu) { (void)gpu; return "n/a"; }
  virtual gm_status_t peer_link(uint32_t gpu, uint32_t peer, PeerLink* out) = 0;
  // Every physical xGMI link of the GPU, in the index order of the per-link telemetry fields.
  virtual gm_status_t xgmi_links(uint32_t gpu, std::vector<XgmiLink>* out) {
    (void)gpu, (void)out;
    return GM_ST_NOT_SUPPORTED;
  }
  // Waits up to timeout_ms for driver events; returns GM_ST_NOT_SUPPORTED when
  // the backend has no event channel (the engine then stops polling).
  virtual gm_status_t poll_events(int timeout_ms, std::vector<DeviceEvent>* out) = 0;
  // True when poll_events itself reports RAS errors (from the driver's error records); the engine then does not
  // derive them from steps of the error counters, which would say everything twice.
  virtual bool delivers_ras_events() { return false; }
  // Backend-specific control channel (fake backend scripting).
  virtual gm_status_t control(const std::string& command) { (void)command; return GM_ST_NOT_SUPPORTED; }
  // Re-enumerate after a GPU reset.
  virtual void refresh_handles() {}
  // Version of the user-space stack under the backend: the ROCm release when it can be found, else the
  // device library's own version (the counterpart of the reference's CUDA driver version,
  // bindings/go/nvml/bindings.go:152-165).  `text` is the full string, e.g. "7.2.0" or "amdsmi 26.2.0".
  virtual gm_status_t runtime_version(uint32_t* major, uint32_t* minor, std::string* text) {
    (void)major, (void)minor, (void)text;
    return GM_ST_NOT_SUPPORTED;
  }
};

std::unique_ptr<Backend> make_fake_backend(const std::string& spec);
std::unique_ptr<Backend> make_amdsmi_backend(const std::string& library_path);
std::unique_ptr<Backend> make_replicate_backend(std::unique_ptr<Backend> inner, uint32_t times);
// "amdsmi" | "fake"; nullptr on an unknown name.
std::unique_ptr<Backend> make_backend(const std::string& name, const std::string& spec);

}  // namespace gpumon
