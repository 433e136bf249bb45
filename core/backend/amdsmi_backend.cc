// AmdSmiBackend: the real device layer.  libamd_smi.so is loaded at run time
// through an explicit function-pointer table, so every gpumon binary builds and
// starts on hosts without ROCm and reports "library not found" instead of
// failing to link — the job of the reference's dlopen shim
// (bindings/go/nvml/nvml_dl.c:8-46), done with a full table instead of its
// ignore-unresolved-symbols link trick (bindings/go/nvml/bindings.go:5).
// Symbols that newer/older library builds may lack are optional: a missing one
// blanks the fields it feeds (GM_VS_NOT_SUPPORTED), nothing else.
//
// Cost model: one amdsmi_get_gpu_metrics_info call per GPU per tick feeds the
// METRICS, VIOLATION, ENERGY and PROF sources (≈80 % of all fields); VRAM, ECC,
// bad pages and the performance level are separate sysfs-backed calls that the
// engine can watch at their own, slower, frequencies.
#include <algorithm>
#include <atomic>
#include <climits>
#include <cstdlib>
#include <cstring>
#include <dlfcn.h>
#include <fcntl.h>
#include <map>
#include <mutex>
#include <thread>
#include <unistd.h>

#include <amd_smi/amdsmi.h>

#include "../common/util.h"
#include "../prof/prof_child.h"
#include "../prof/prof_counters.h"
#include "backend.h"
#include "metrics_table.h"
#include "procscan.h"

namespace gpumon {
namespace {

// X(name, required)
#define AMDSMI_FUNCS(X)                               \
  X(amdsmi_init, 1)                                   \
  X(amdsmi_shut_down, 1)                              \
  X(amdsmi_get_socket_handles, 1)                     \
  X(amdsmi_get_processor_handles, 1)                  \
  X(amdsmi_get_processor_type, 1)                     \
  X(amdsmi_get_gpu_device_bdf, 1)                     \
  X(amdsmi_get_gpu_device_uuid, 1)                    \
  X(amdsmi_get_gpu_metrics_info, 1)                   \
  X(amdsmi_status_code_to_string, 0)                  \
  X(amdsmi_get_gpu_enumeration_info, 0)               \
  X(amdsmi_get_gpu_asic_info, 0)                      \
  X(amdsmi_get_gpu_board_info, 0)                     \
  X(amdsmi_get_gpu_vbios_info, 0)                     \
  X(amdsmi_get_gpu_driver_info, 0)                    \
  X(amdsmi_get_lib_version, 0)                        \
  X(amdsmi_get_fw_info, 0)                            \
  X(amdsmi_get_gpu_kfd_info, 0)                       \
  X(amdsmi_get_power_cap_info, 0)                     \
  X(amdsmi_get_gpu_memory_total, 0)                   \
  X(amdsmi_get_gpu_memory_usage, 0)                   \
  X(amdsmi_get_pcie_info, 0)                          \
  X(amdsmi_get_clock_info, 0)                         \
  X(amdsmi_get_energy_count, 0)                       \
  X(amdsmi_get_gpu_total_ecc_count, 0)                \
  X(amdsmi_get_gpu_ecc_count, 0)                      \
  X(amdsmi_get_gpu_bad_page_info, 0)                  \
  X(amdsmi_get_gpu_perf_level, 0)                     \
  X(amdsmi_get_gpu_process_list, 0)                   \
  X(amdsmi_get_gpu_compute_partition, 0)              \
  X(amdsmi_get_gpu_memory_partition, 0)               \
  X(amdsmi_topo_get_link_type, 0)                     \
  X(amdsmi_topo_get_link_weight, 0)                   \
  X(amdsmi_topo_get_numa_node_number, 0)              \
  X(amdsmi_get_minmax_bandwidth_between_processors, 0)\
  X(amdsmi_is_P2P_accessible, 0)                      \
  X(amdsmi_init_gpu_event_notification, 0)            \
  X(amdsmi_set_gpu_event_notification_mask, 0)        \
  X(amdsmi_get_gpu_event_notification, 0)             \
  X(amdsmi_stop_gpu_event_notification, 0)            \
  X(amdsmi_get_link_metrics, 0)                        \
  X(amdsmi_get_gpu_xgmi_link_status, 0)                \
  X(amdsmi_gpu_xgmi_error_status, 0)                   \
  X(amdsmi_get_gpu_pci_throughput, 0)                  \
  X(amdsmi_get_violation_status, 0)                    \
  X(amdsmi_get_gpu_cper_entries, 0)

struct SmiApi {
  void* lib = nullptr;
#define X(name, req) decltype(&::name) name = nullptr;
  AMDSMI_FUNCS(X)
#undef X
};

gm_status_t load_api(const std::string& hint, SmiApi* api, std::string* err) {
  std::vector<std::string> candidates;
  if (!hint.empty()) candidates.push_back(hint);
  const char* env = getenv("GPUMON_AMDSMI_LIB");
  if (env && *env) candidates.push_back(env);
  candidates.push_back("libamd_smi.so");
  const char* rocm = getenv("ROCM_PATH");
  candidates.push_back(std::string(rocm && *rocm ? rocm : "/opt/rocm") + "/lib/libamd_smi.so");
  for (auto& c : candidates) {
    api->lib = dlopen(c.c_str(), RTLD_LAZY | RTLD_LOCAL);
    if (api->lib) break;
    *err = dlerror();
  }
  if (!api->lib) return GM_ST_LIBRARY_NOT_FOUND;
#define X(name, req)                                                       \
  api->name = reinterpret_cast<decltype(api->name)>(dlsym(api->lib, #name)); \
  if (!api->name && req) {                                                 \
    *err = "missing symbol " #name;                                        \
    dlclose(api->lib);                                                     \
    api->lib = nullptr;                                                    \
    return GM_ST_FUNCTION_NOT_FOUND;                                       \
  }
  AMDSMI_FUNCS(X)
#undef X
  return GM_ST_OK;
}

uint8_t blank_reason(amdsmi_status_t st) {
  switch (st) {
    case AMDSMI_STATUS_NOT_SUPPORTED:
    case AMDSMI_STATUS_NOT_YET_IMPLEMENTED:
    case AMDSMI_STATUS_NO_DATA:
    case AMDSMI_STATUS_FILE_NOT_FOUND:
    case AMDSMI_STATUS_SETTING_UNAVAILABLE:
      return GM_VS_NOT_SUPPORTED;
    case AMDSMI_STATUS_NO_PERM:
      return GM_VS_NO_PERMISSION;
    default:
      return GM_VS_ERROR;
  }
}

constexpr int64_t kTableRecheckUs = 300ll * 1000000;  // the direct metrics-table decoding is held against the library's this often

template <typename T>
bool valid(T v) {
  return v != (T)~(T)0;
}

static_assert(kMtXcp == AMDSMI_MAX_NUM_XCP && kMtXcc == AMDSMI_MAX_NUM_XCC && kMtHbm == AMDSMI_NUM_HBM_INSTANCES &&
                  kMtVcn == AMDSMI_MAX_NUM_VCN && kMtGfxClks == AMDSMI_MAX_NUM_GFX_CLKS && kMtClks == AMDSMI_MAX_NUM_CLKS &&
                  kMtLinks == AMDSMI_MAX_NUM_XGMI_LINKS,
              "MetricsTable mirrors the array bounds of amdsmi_gpu_metrics_t");

// The members the sampler consumes, out of the library's all-versions struct.
void from_library(const amdsmi_gpu_metrics_t& a, MetricsTable* m) {
  metrics_table_clear(m);
#define CP(name) m->name = a.name
  CP(temperature_edge); CP(temperature_hotspot); CP(temperature_mem); CP(temperature_vrgfx); CP(temperature_vrsoc); CP(temperature_vrmem);
  CP(average_gfx_activity); CP(average_umc_activity); CP(average_mm_activity); CP(average_socket_power); CP(current_socket_power);
  CP(current_gfxclk); CP(current_socclk); CP(current_uclk); CP(current_vclk0); CP(current_dclk0);
  CP(pcie_link_width); CP(pcie_link_speed); CP(xgmi_link_width); CP(xgmi_link_speed); CP(num_partition);
  CP(pcie_nak_sent_count_acc); CP(pcie_nak_rcvd_count_acc); CP(energy_accumulator); CP(firmware_timestamp);
  CP(pcie_bandwidth_inst); CP(pcie_l0_to_recov_count_acc); CP(pcie_replay_count_acc); CP(accumulation_counter);
  CP(prochot_residency_acc); CP(ppt_residency_acc); CP(socket_thm_residency_acc); CP(vr_thm_residency_acc); CP(hbm_thm_residency_acc);
#undef CP
#define CPA(name) memcpy(m->name, a.name, sizeof m->name); static_assert(sizeof m->name == sizeof a.name, #name)
  CPA(temperature_hbm); CPA(current_gfxclks); CPA(current_socclks); CPA(current_vclk0s); CPA(current_dclk0s); CPA(vcn_activity);
  CPA(xgmi_link_status); CPA(xgmi_read_data_acc); CPA(xgmi_write_data_acc);
#undef CPA
  for (int p = 0; p < kMtXcp; p++)
    for (int x = 0; x < kMtXcc; x++) {
      m->xcp_stats[p].gfx_busy_inst[x] = a.xcp_stats[p].gfx_busy_inst[x];
      m->xcp_stats[p].gfx_below_host_limit_acc[x] = a.xcp_stats[p].gfx_below_host_limit_acc[x];
      m->xcp_stats[p].gfx_below_host_limit_total_acc[x] = a.xcp_stats[p].gfx_below_host_limit_total_acc[x];
      m->xcp_stats[p].gfx_low_utilization_acc[x] = a.xcp_stats[p].gfx_low_utilization_acc[x];
    }
}

struct Dev {
  amdsmi_processor_handle h = nullptr;
  GpuStatic st;
  uint64_t bdf_key = 0;
  uint32_t partition_id = 0;
  ProcSource proc_source = ProcSource::NONE;  // where the last process list came from
  double energy_resolution_uj = 15.259;  // µJ per accumulator tick, refined at init
  // throttle-residency calibration and edge detection
  uint64_t prev_acc_counter = 0, prev_fw_ts = 0;
  double acc_tick_us = 1000.0;  // PMFW accumulates once per millisecond until measured otherwise
  uint64_t prev_res[5] = {0, 0, 0, 0, 0};
  uint64_t prev_low = 0, prev_host = 0;  // low-utilisation / below-host-limit residency, worst XCC
  struct Residency {  // one exported duration: power, thermal, board, low utilisation, host limit
    bool seen = false;
    uint64_t ticks = 0;
    double us = 0;
  };
  Residency residency[5];
  bool have_prev = false;
  bool events_on = false;
  // RAS records (CPER) cached by the driver: read position, and whether this user may read them at all
  uint64_t cper_cursor = 0;
  bool cper_ok = true, cper_primed = false;
  bool violation_checked = false;
  // Direct reads of the driver's metrics table (metrics_table.h): the kept descriptor, -1 while the library reads.
  int table_fd = -1;
  int64_t table_checked_mono = 0;   // last comparison with the library's decoding
  std::string table_note;           // why direct reads are off, for metrics_table_source()
  // CU occupancy comes from the KFD process table, which costs a walk over /proc: read it at most once a
  // second however fast the profiling fields are watched, and not at all where the table is not readable.
  int64_t occupancy_read_mono = 0;
  double occupancy = std::nan("");
  // Counter plugin (libgpumon_prof.so): agent of this device, last window's MFMA activity.
  int64_t throttle_bits = 0;
  int prof_agent = -1;
  int64_t prof_read_mono = 0;
  // Last window of each counter-derived value; the in-process plugin measures one GRBM signal per window, so
  // every value carries the time it was taken and goes blank when it is older than a few windows.
  struct ProfValue {
    double v = std::nan("");
    int64_t mono = 0;
  };
  ProfValue mfma_active, cu_active, mem_pipe_active, gpu_active_ctr;
};

// Entry points of the optional counter plugin (core/prof/prof_counters.h), resolved at run time.
struct ProfApi {
  void* lib = nullptr;
  decltype(&::gmp_open) open = nullptr;
  decltype(&::gmp_agent_info) agent_info = nullptr;
  decltype(&::gmp_sample) sample = nullptr;
  decltype(&::gmp_last_error) last_error = nullptr;
  decltype(&::gmp_close) close = nullptr;
  decltype(&::gmp_abi_version) abi_version = nullptr;
  decltype(&::gmp_set_count) set_count = nullptr;
};

// "/sys", or the tree a test has put in its place.
const std::string& sysfs_root() {
  static const std::string root = [] {
    const char* e = getenv("GPUMON_SYSFS_ROOT");
    return std::string(e && *e ? e : "/sys");
  }();
  return root;
}

std::string sysfs_trim(const std::string& path) {
  std::string s;
  return read_file(path, &s) ? trim(s) : std::string();
}

int pcie_gen_from_speed_dgts(uint32_t tenth_gts) {  // 0.1 GT/s units
  if (tenth_gts >= 640) return 6;
  if (tenth_gts >= 320) return 5;
  if (tenth_gts >= 160) return 4;
  if (tenth_gts >= 80) return 3;
  if (tenth_gts >= 50) return 2;
  if (tenth_gts >= 25) return 1;
  return 0;
}

// How two PCIe functions relate, from their sysfs device paths.
gm_link_type_t classify_pcie(const std::string& bdf_a, const std::string& bdf_b, int numa_a, int numa_b) {
  char pa[PATH_MAX], pb[PATH_MAX];
  std::string a = sysfs_root() + "/bus/pci/devices/" + bdf_a, b = sysfs_root() + "/bus/pci/devices/" + bdf_b;
  if (!realpath(a.c_str(), pa) || !realpath(b.c_str(), pb))
    return numa_a >= 0 && numa_a == numa_b ? GM_LINK_SAME_CPU : GM_LINK_CROSS_CPU;
  std::vector<std::string> ca = split(pa, '/'), cb = split(pb, '/');
  size_t common = 0;
  while (common < ca.size() && common < cb.size() && ca[common] == cb[common]) common++;
  // components: "", "sys", "devices", "pciDDDD:BB", bridge..., device
  size_t root = 0;
  for (size_t i = 0; i < ca.size(); i++)
    if (ca[i].compare(0, 3, "pci") == 0) { root = i; break; }
  if (!root || common <= root) return numa_a >= 0 && numa_a == numa_b ? GM_LINK_SAME_CPU : GM_LINK_CROSS_CPU;
  size_t shared_bridges = common - root - 1;  // bridges below the root complex both paths share
  size_t depth_a = ca.size() - root - 2, depth_b = cb.size() - root - 2;  // bridges above each device
  if (shared_bridges == 0) return GM_LINK_HOST_BRIDGE;
  if (shared_bridges >= depth_a && shared_bridges >= depth_b) return GM_LINK_SINGLE_SWITCH;
  return GM_LINK_MULTI_SWITCH;
}

class AmdSmiBackend : public Backend {
 public:
  explicit AmdSmiBackend(const std::string& lib) : lib_hint_(lib) {}
  ~AmdSmiBackend() override { shutdown(); }
  const char* name() const override { return "amdsmi"; }

  gm_status_t init() override {
    std::lock_guard<std::mutex> l(mu_);
    std::string err;
    gm_status_t st = load_api(lib_hint_, &api_, &err);
    if (st != GM_ST_OK) {
      GM_LOGE("amdsmi: %s", err.c_str());
      return st;
    }
    amdsmi_status_t rc = api_.amdsmi_init(AMDSMI_INIT_AMD_GPUS);
    if (rc != AMDSMI_STATUS_SUCCESS) {
      GM_LOGE("amdsmi_init failed: %s", errstr(rc));
      dlclose(api_.lib);
      api_.lib = nullptr;
      return rc == AMDSMI_STATUS_NO_PERM ? GM_ST_NO_PERMISSION : GM_ST_INIT_ERROR;
    }
    inited_ = true;
    gm_status_t est = enumerate();
    if (est == GM_ST_OK && api_.amdsmi_get_gpu_pci_throughput && !devs_.empty()) {
      pcie_run_ = true;
      pcie_thread_ = std::thread([this] { pcie_loop(); });
    }
    return est;
  }

  void shutdown() override {
    pcie_run_ = false;
    if (pcie_thread_.joinable()) pcie_thread_.join();  // before mu_: the loop takes it once a round
    {
      std::vector<std::thread> checks;
      {
        std::lock_guard<std::mutex> cl(checks_mu_);
        checks.swap(checks_);
      }
      for (auto& t : checks)
        if (t.joinable()) t.join();  // none of them takes mu_; the library must outlive them
    }
    std::lock_guard<std::mutex> l(mu_);
    if (!inited_) return;
    for (auto& d : devs_)
      if (d.events_on && api_.amdsmi_stop_gpu_event_notification) api_.amdsmi_stop_gpu_event_notification(d.h);
    api_.amdsmi_shut_down();
    if (prof_.close) prof_.close();
    prof_child_.reset();
    prof_tried_ = false;  // a later init maps the plugin's agents onto the new device table
    inited_ = false;
    close_tables(devs_);
    devs_.clear();
    // The library stays mapped: amdsmi keeps thread-local state that outlives shut_down.
  }

  uint32_t device_count() override { return (uint32_t)devs_.size(); }

  gm_status_t static_info(uint32_t gpu, GpuStatic* out) override {
    if (gpu >= devs_.size()) return GM_ST_BADPARAM;
    *out = devs_[gpu].st;
    return GM_ST_OK;
  }

  void refresh_handles() override {
    std::lock_guard<std::mutex> l(mu_);
    if (inited_) enumerate();
  }

  // The ROCm release that the loaded device library belongs to (<prefix>/.info/version next to its lib
  // directory), else the library's own version.
  gm_status_t runtime_version(uint32_t* major, uint32_t* minor, std::string* text) override {
    std::string v;
    Dl_info info;
    if (api_.amdsmi_init && dladdr(reinterpret_cast<void*>(api_.amdsmi_init), &info) && info.dli_fname) {
      char real[PATH_MAX];
      std::string path = realpath(info.dli_fname, real) ? real : info.dli_fname;
      const size_t file = path.rfind('/');  // <prefix>/lib/libamd_smi.so.N -> <prefix>
      const size_t dir = file != std::string::npos && file > 0 ? path.rfind('/', file - 1) : std::string::npos;
      if (dir != std::string::npos && read_file(path.substr(0, dir) + "/.info/version", &v)) v = trim(v);
      else v.clear();
    }
    unsigned a = 0, b = 0;
    if (!v.empty() && sscanf(v.c_str(), "%u.%u", &a, &b) == 2) {
      *major = a, *minor = b, *text = v;
      return GM_ST_OK;
    }
    if (!api_.amdsmi_get_lib_version) return GM_ST_NOT_SUPPORTED;
    amdsmi_version_t lv;
    memset(&lv, 0, sizeof lv);
    if (api_.amdsmi_get_lib_version(&lv) != AMDSMI_STATUS_SUCCESS) return GM_ST_NOT_SUPPORTED;
    *major = lv.major, *minor = lv.minor;
    *text = strprintf("amdsmi %u.%u.%u", lv.major, lv.minor, lv.release);
    return GM_ST_OK;
  }

  gm_status_t sample(uint32_t gpu, uint32_t sources, Sample* out) override {
    if (gpu >= devs_.size()) return GM_ST_BADPARAM;
    Dev& d = devs_[gpu];
    bool any_ok = false;
    if (sources & (SRC_METRICS | SRC_VIOLATION | SRC_ENERGY | SRC_PROF)) any_ok |= sample_metrics(d, sources, out);
    if (sources & SRC_VRAM) any_ok |= sample_vram(d, out);
    if (sources & SRC_ECC) any_ok |= sample_ecc(d, out);
    if (sources & SRC_BAD_PAGES) any_ok |= sample_bad_pages(d, out);
    if (sources & SRC_PERF) any_ok |= sample_perf(d, out);
    return any_ok ? GM_ST_OK : GM_ST_GENERIC_ERROR;
  }

  std::string metrics_table_source(uint32_t gpu) override {
    if (gpu >= devs_.size()) return "n/a";
    const Dev& d = devs_[gpu];
    return d.table_fd >= 0 ? "direct" : d.table_note.empty() ? "library" : "library: " + d.table_note;
  }

  const char* process_source(uint32_t gpu) override {
    return gpu < devs_.size() ? proc_source_name(devs_[gpu].proc_source) : "none";
  }

  gm_status_t processes(uint32_t gpu, std::vector<ProcSample>* out) override {
    out->clear();
    if (gpu >= devs_.size()) return GM_ST_BADPARAM;
    // The library builds its list from the KFD process table.  Where that directory is missing or closed
    // (containers with a partial sysfs, guests, non-root users on some hosts) it reports an EMPTY list as success
    // and complains on stderr at every call.  There the kernel's per-file DRM statistics under /proc answer
    // instead (core/backend/procscan.h): every process this one may look into, with memory and engine time where
    // the kernel writes them and as a bare holder of the render node where it does not.
    static const int kfd_errno = access((sysfs_root() + "/class/kfd/kfd/proc").c_str(), R_OK | X_OK) == 0 ? 0 : errno;
    if (kfd_errno || !api_.amdsmi_get_gpu_process_list) {
      static const std::string proc_root = getenv("GPUMON_PROC_ROOT") ? getenv("GPUMON_PROC_ROOT") : "/proc";
      ProcScan scan;
      scan_gpu_processes(proc_root, devs_[gpu].st.pci_bdf, devs_[gpu].st.drm_render_minor, &scan);
      devs_[gpu].proc_source = scan.source;
      if (scan.source == ProcSource::NONE) return kfd_errno == EACCES ? GM_ST_NO_PERMISSION : GM_ST_NOT_SUPPORTED;
      *out = std::move(scan.procs);
      return GM_ST_OK;
    }
    devs_[gpu].proc_source = ProcSource::KFD;
    uint32_t n = 64;
    std::vector<amdsmi_proc_info_t> list;
    amdsmi_status_t rc = AMDSMI_STATUS_SUCCESS;
    for (int attempt = 0; attempt < 3; attempt++) {
      list.assign(n, amdsmi_proc_info_t());
      uint32_t cap = n;
      rc = api_.amdsmi_get_gpu_process_list(devs_[gpu].h, &n, list.data());
      if (rc == AMDSMI_STATUS_OUT_OF_RESOURCES && n > cap) continue;  // n now holds the real count
      break;
    }
    if (rc != AMDSMI_STATUS_SUCCESS && rc != AMDSMI_STATUS_OUT_OF_RESOURCES)
      return rc == AMDSMI_STATUS_NO_PERM ? GM_ST_NO_PERMISSION
                                         : (rc == AMDSMI_STATUS_NOT_SUPPORTED ? GM_ST_NOT_SUPPORTED : GM_ST_GENERIC_ERROR);
    for (uint32_t i = 0; i < n && i < list.size(); i++) {
      ProcSample p;
      p.pid = (uint32_t)list[i].pid;
      p.name = list[i].name;
      p.vram_bytes = list[i].memory_usage.vram_mem ? list[i].memory_usage.vram_mem : list[i].mem;
      p.gtt_bytes = list[i].memory_usage.gtt_mem;
      p.cpu_mem_bytes = list[i].memory_usage.cpu_mem;
      p.gfx_engine_ns = list[i].engine_usage.gfx;
      p.enc_engine_ns = list[i].engine_usage.enc;
      p.cu_occupancy = list[i].cu_occupancy;
      out->push_back(p);
    }
    return GM_ST_OK;
  }

  gm_status_t peer_link(uint32_t gpu, uint32_t peer, PeerLink* out) override {
    if (gpu >= devs_.size() || peer >= devs_.size() || gpu == peer) return GM_ST_BADPARAM;
    if (!api_.amdsmi_topo_get_link_type) return GM_ST_NOT_SUPPORTED;
    Dev& a = devs_[gpu];
    Dev& b = devs_[peer];
    uint64_t hops = 0;
    amdsmi_link_type_t type = AMDSMI_LINK_TYPE_UNKNOWN;
    amdsmi_status_t rc = api_.amdsmi_topo_get_link_type(a.h, b.h, &hops, &type);
    if (rc == AMDSMI_STATUS_NOT_SUPPORTED) return GM_ST_NOT_SUPPORTED;
    if (rc != AMDSMI_STATUS_SUCCESS) return GM_ST_GENERIC_ERROR;
    PeerLink p;
    p.hops = (uint32_t)hops;
    if (type == AMDSMI_LINK_TYPE_XGMI) {
      p.link = GM_LINK_XGMI;
      // Parallel links to this peer = physical links whose far end carries the peer's PCI address (the count the
      // reference takes from the NVLink remote PCI info, bindings/go/nvml/nvml.go:539-568).  Where the platform
      // hides the far end (virtualised hosts report an all-ones address) an xGMI neighbour has at least one.
      std::vector<XgmiLink> links;
      uint32_t to_peer = 0;
      if (xgmi_links_locked(a, &links) == GM_ST_OK)
        for (auto& l : links)
          if (l.state != 0 && !l.peer_bdf.empty() && l.peer_bdf == b.st.pci_bdf) to_peer++;
      p.num_links = to_peer ? to_peer : 1;
    } else if (type == AMDSMI_LINK_TYPE_PCIE) {
      p.link = classify_pcie(a.st.pci_bdf, b.st.pci_bdf, a.st.numa_node, b.st.numa_node);
    } else if (type == AMDSMI_LINK_TYPE_INTERNAL) {
      p.link = GM_LINK_SAME_BOARD;
    }
    if (api_.amdsmi_topo_get_link_weight) api_.amdsmi_topo_get_link_weight(a.h, b.h, &p.weight);
    if (api_.amdsmi_get_minmax_bandwidth_between_processors) {
      uint64_t mn = 0, mx = 0;
      if (api_.amdsmi_get_minmax_bandwidth_between_processors(a.h, b.h, &mn, &mx) == AMDSMI_STATUS_SUCCESS) {
        p.min_bw_mbs = (int64_t)mn;
        p.max_bw_mbs = (int64_t)mx;
      }
    }
    if (api_.amdsmi_is_P2P_accessible) {
      bool acc = false;
      if (api_.amdsmi_is_P2P_accessible(a.h, b.h, &acc) == AMDSMI_STATUS_SUCCESS) p.p2p_accessible = acc;
    }
    *out = p;
    return GM_ST_OK;
  }

  bool delivers_ras_events() override { return cper_any_; }

  gm_status_t xgmi_links(uint32_t gpu, std::vector<XgmiLink>* out) override {
    out->clear();
    if (gpu >= devs_.size()) return GM_ST_BADPARAM;
    return xgmi_links_locked(devs_[gpu], out);
  }

  // Link table from amdsmi_get_link_metrics (far end, rate, traffic) joined with the per-link state; on libraries
  // without it, state and traffic from the metrics table and no far end.
  gm_status_t xgmi_links_locked(Dev& d, std::vector<XgmiLink>* out) {
    out->clear();
    int state[GM_MAX_XGMI_LINKS];
    for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) state[i] = -1;
    bool have_state = false;
    if (api_.amdsmi_get_gpu_xgmi_link_status) {
      amdsmi_xgmi_link_status_t ls;
      memset(&ls, 0, sizeof ls);
      if (api_.amdsmi_get_gpu_xgmi_link_status(d.h, &ls) == AMDSMI_STATUS_SUCCESS) {
        have_state = true;
        for (uint32_t i = 0; i < ls.total_links && i < GM_MAX_XGMI_LINKS && i < AMDSMI_MAX_NUM_XGMI_LINKS; i++)
          state[i] = ls.status[i] == AMDSMI_XGMI_LINK_UP ? 1 : ls.status[i] == AMDSMI_XGMI_LINK_DOWN ? 0 : -1;
      }
    }
    amdsmi_gpu_metrics_t m;
    memset(&m, 0xff, sizeof m);
    const bool have_metrics = api_.amdsmi_get_gpu_metrics_info(d.h, &m) == AMDSMI_STATUS_SUCCESS;
    if (!have_state && have_metrics)
      for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) {
        const uint16_t ls = m.xgmi_link_status[i];
        if (valid(ls) && ls != AMDSMI_XGMI_LINK_DISABLE) state[i] = ls == AMDSMI_XGMI_LINK_UP ? 1 : 0, have_state = true;
      }
    std::unique_ptr<amdsmi_link_metrics_t> lm;
    if (api_.amdsmi_get_link_metrics) {
      lm.reset(new amdsmi_link_metrics_t);
      memset(lm.get(), 0, sizeof *lm);
      if (api_.amdsmi_get_link_metrics(d.h, lm.get()) != AMDSMI_STATUS_SUCCESS) lm.reset();
    }
    if (!have_state && !lm) return GM_ST_NOT_SUPPORTED;
    for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) {
      XgmiLink l;
      l.index = (uint32_t)i;
      l.state = state[i];
      if (have_metrics) {
        if (valid(m.xgmi_read_data_acc[i])) l.rx_kb = (int64_t)m.xgmi_read_data_acc[i];
        if (valid(m.xgmi_write_data_acc[i])) l.tx_kb = (int64_t)m.xgmi_write_data_acc[i];
      }
      // Some library builds leave num_links at 0 and still fill the entries: go by the entry's link type.
      if (lm && (uint32_t)i < AMDSMI_MAX_NUM_XGMI_PHYSICAL_LINK && lm->links[i].link_type == AMDSMI_LINK_TYPE_XGMI) {
        const auto& e = lm->links[i];
        if (e.bit_rate) l.bit_rate_gbps = e.bit_rate;
        if (e.max_bandwidth) l.max_bandwidth_gbps = e.max_bandwidth;
        if (l.rx_kb < 0) l.rx_kb = (int64_t)e.read;
        if (l.tx_kb < 0) l.tx_kb = (int64_t)e.write;
        // An all-ones address is the library's "unknown" (seen on virtualised single-GPU hosts).
        if (e.bdf.as_uint != ~0ull && e.bdf.as_uint != 0)
          l.peer_bdf = strprintf("%04llx:%02x:%02x.%x", (unsigned long long)e.bdf.domain_number, (unsigned)e.bdf.bus_number,
                                 (unsigned)e.bdf.device_number, (unsigned)e.bdf.function_number);
      }
      if (l.state < 0 && l.peer_bdf.empty() && l.rx_kb <= 0 && l.tx_kb <= 0) continue;  // not a wired link
      out->push_back(l);
    }
    return GM_ST_OK;
  }

  // One-time cross-check of the residency accumulators this backend reads from the metrics table against the
  // library's own violation report (amdsmi_get_violation_status).  That call sleeps 100 ms between two reads of
  // its own, so it runs once, on a thread of its own, never in the sampler.  shutdown() joins the thread before it
  // closes the library: left to itself (it was detached once) it could still be inside the library, asleep, when a
  // short-lived caller shut the library down under it, and the heap paid for it a moment later.
  void check_violation_source(Dev& d) {
    if (!api_.amdsmi_get_violation_status) return;
    amdsmi_processor_handle h = d.h;
    const uint32_t index = d.st.index;
    auto fn_status = api_.amdsmi_get_violation_status;
    auto fn_metrics = api_.amdsmi_get_gpu_metrics_info;
    std::lock_guard<std::mutex> cl(checks_mu_);
    checks_.emplace_back([h, index, fn_status, fn_metrics] {
      std::unique_ptr<amdsmi_violation_status_t> v(new amdsmi_violation_status_t);
      memset(v.get(), 0xff, sizeof *v);
      if (fn_status(h, v.get()) != AMDSMI_STATUS_SUCCESS) return;
      amdsmi_gpu_metrics_t m;
      memset(&m, 0xff, sizeof m);
      if (fn_metrics(h, &m) != AMDSMI_STATUS_SUCCESS) return;
      auto close_to = [](uint64_t a, uint64_t b) { return a <= b && b - a <= b / 100 + 2000; };  // the table was read later
      const bool counter_ok = !valid(v->acc_counter) || !valid(m.accumulation_counter) || close_to(v->acc_counter, m.accumulation_counter);
      const bool ppt_ok = !valid(v->acc_ppt_pwr) || !valid(m.ppt_residency_acc) || close_to(v->acc_ppt_pwr, m.ppt_residency_acc);
      if (counter_ok && ppt_ok) GM_LOGI("amdsmi: GPU %u residency accumulators agree with the library's violation report", index);
      else GM_LOGW("amdsmi: GPU %u residency accumulators differ from the library's violation report (counter %llu vs %llu, ppt %llu vs %llu)",
                   index, (unsigned long long)v->acc_counter, (unsigned long long)m.accumulation_counter,
                   (unsigned long long)v->acc_ppt_pwr, (unsigned long long)m.ppt_residency_acc);
    });
  }
  std::mutex checks_mu_;
  std::vector<std::thread> checks_;  // one-time cross-checks that call the library from threads of their own

  // ---- PCIe TX / RX ---------------------------------------------------------------
  // amdsmi_get_gpu_pci_throughput measures for one second inside the driver, so it cannot sit in the sampler.  A
  // thread of its own asks every GPU in turn, at most every five seconds, and integrates the measured rates into
  // two cumulative KB counters per GPU (what the contract metrics are).  Hosts that do not offer the measurement
  // answer the first round with "not supported" at once; the thread then ends and the fields stay blank.
  struct PcieAcc {
    double tx_kb = 0, rx_kb = 0;
    int64_t last_mono = 0;
    uint64_t readings = 0;
  };
  void pcie_loop() {
    while (pcie_run_) {
      std::vector<std::pair<amdsmi_processor_handle, uint64_t>> targets;
      {
        std::lock_guard<std::mutex> l(mu_);
        if (!inited_) break;
        for (auto& d : devs_) targets.push_back({d.h, d.bdf_key});
      }
      const int64_t pass_t0 = monotonic_us();
      bool any = false;
      for (auto& t : targets) {
        if (!pcie_run_) break;
        uint64_t sent = 0, received = 0, max_pkt = 0;
        if (api_.amdsmi_get_gpu_pci_throughput(t.first, &sent, &received, &max_pkt) != AMDSMI_STATUS_SUCCESS) continue;
        any = true;
        const int64_t now = monotonic_us();
        std::lock_guard<std::mutex> pl(pcie_mu_);
        PcieAcc& a = pcie_acc_[t.second];
        const double seconds = a.last_mono ? std::min(60.0, (double)(now - a.last_mono) / 1e6) : 1.0;
        a.tx_kb += (double)sent / 1024.0 * seconds;
        a.rx_kb += (double)received / 1024.0 * seconds;
        a.last_mono = now;
        a.readings++;
      }
      if (!any) {
        GM_LOGI("amdsmi: PCIe throughput is not measured on this host; dcgm_pcie_tx/rx_throughput stay blank");
        break;
      }
      for (int64_t left = 5000000 - (monotonic_us() - pass_t0); left > 0 && pcie_run_; left -= 100000) usleep(100000);
    }
  }

  // ---- RAS records -----------------------------------------------------------------
  // New CPER entries of fatal / uncorrected severity become RAS events.  Entries that were already in the driver's
  // cache when monitoring started are skipped: they are history, not news.
  void poll_cper(std::vector<DeviceEvent>* out) {
    if (!api_.amdsmi_get_gpu_cper_entries) return;
    const int64_t now = monotonic_us();
    if (cper_read_mono_ && now - cper_read_mono_ < 5000000) return;
    cper_read_mono_ = now;
    std::lock_guard<std::mutex> l(mu_);
    if (!inited_) return;
    std::vector<char> buf(64 << 10);
    amdsmi_cper_hdr_t* hdrs[64];
    bool any_ok = false;
    for (auto& d : devs_) {
      if (!d.cper_ok) continue;
      for (int round = 0; round < 16; round++) {
        uint64_t size = buf.size(), count = 64;
        const uint32_t mask = (1u << AMDSMI_CPER_SEV_FATAL) | (1u << AMDSMI_CPER_SEV_NON_FATAL_UNCORRECTED);
        amdsmi_status_t rc = api_.amdsmi_get_gpu_cper_entries(d.h, mask, buf.data(), &size, hdrs, &count, &d.cper_cursor);
        if (rc != AMDSMI_STATUS_SUCCESS && rc != AMDSMI_STATUS_MORE_DATA) {
          if (rc == AMDSMI_STATUS_NO_PERM || rc == AMDSMI_STATUS_NOT_SUPPORTED || rc == AMDSMI_STATUS_NOT_YET_IMPLEMENTED ||
              rc == AMDSMI_STATUS_FILE_NOT_FOUND) {
            d.cper_ok = false;
            GM_LOGI("amdsmi: RAS records (CPER) of GPU %u are not readable (%s); RAS events come from counter steps", d.st.index, errstr(rc));
          }
          break;
        }
        any_ok = true;
        if (d.cper_primed)
          for (uint64_t i = 0; i < count && i < 64; i++) {
            DeviceEvent e;
            e.gpu = d.st.index;
            e.ts_us = now_us();
            e.code = hdrs[i]->error_severity == AMDSMI_CPER_SEV_FATAL ? GM_EVENT_RAS_UE : GM_EVENT_RAS_POISON;
            e.message = strprintf("RAS record %.8s: %s error", hdrs[i]->record_id,
                                  hdrs[i]->error_severity == AMDSMI_CPER_SEV_FATAL ? "fatal" : "uncorrected (deferred)");
            out->push_back(e);
          }
        if (rc != AMDSMI_STATUS_MORE_DATA) break;
      }
      d.cper_primed = true;
    }
    cper_any_ = any_ok;
  }

  gm_status_t poll_events(int timeout_ms, std::vector<DeviceEvent>* out) override {
    const bool have_channel_api = api_.amdsmi_init_gpu_event_notification && api_.amdsmi_get_gpu_event_notification &&
                                  api_.amdsmi_set_gpu_event_notification_mask;
    poll_cper(out);
    {
      std::lock_guard<std::mutex> l(mu_);
      if (!inited_) return GM_ST_UNINITIALIZED;
      if (!events_tried_ && have_channel_api) {
        events_tried_ = true;
        uint64_t mask = AMDSMI_EVENT_MASK_FROM_INDEX(AMDSMI_EVT_NOTIF_VMFAULT) |
                        AMDSMI_EVENT_MASK_FROM_INDEX(AMDSMI_EVT_NOTIF_THERMAL_THROTTLE) |
                        AMDSMI_EVENT_MASK_FROM_INDEX(AMDSMI_EVT_NOTIF_GPU_PRE_RESET) |
                        AMDSMI_EVENT_MASK_FROM_INDEX(AMDSMI_EVT_NOTIF_GPU_POST_RESET);
        for (auto& d : devs_) {
          amdsmi_status_t rc = api_.amdsmi_init_gpu_event_notification(d.h);
          if (rc == AMDSMI_STATUS_SUCCESS) rc = api_.amdsmi_set_gpu_event_notification_mask(d.h, mask);
          d.events_on = rc == AMDSMI_STATUS_SUCCESS;
          if (!d.events_on) GM_LOGI("amdsmi: no event channel for GPU %u: %s", d.st.index, errstr(rc));
          events_any_ = events_any_ || d.events_on;
        }
      }
      if (!events_any_) {
        if (!cper_any_) return GM_ST_NOT_SUPPORTED;  // neither an event channel nor readable RAS records
      }
    }
    if (!events_any_) {  // RAS records only: pace the caller's loop ourselves
      usleep((useconds_t)std::max(1, timeout_ms) * 1000);
      return GM_ST_OK;
    }
    amdsmi_evt_notification_data_t data[16];
    uint32_t n = 16;
    amdsmi_status_t rc = api_.amdsmi_get_gpu_event_notification(timeout_ms, &n, data);
    if (rc == AMDSMI_STATUS_NO_DATA || rc == AMDSMI_STATUS_TIMEOUT) return GM_ST_OK;
    if (rc != AMDSMI_STATUS_SUCCESS && rc != AMDSMI_STATUS_INSUFFICIENT_SIZE) return GM_ST_GENERIC_ERROR;
    int64_t ts = now_us();
    for (uint32_t i = 0; i < n && i < 16; i++) {
      DeviceEvent e;
      e.ts_us = ts;
      e.message = data[i].message;
      for (auto& d : devs_)
        if (d.h == data[i].processor_handle) e.gpu = d.st.index;
      switch (data[i].event) {
        case AMDSMI_EVT_NOTIF_VMFAULT: e.code = GM_EVENT_VMFAULT; break;
        case AMDSMI_EVT_NOTIF_THERMAL_THROTTLE: e.code = GM_EVENT_THERMAL_THROTTLE; break;
        case AMDSMI_EVT_NOTIF_GPU_PRE_RESET: e.code = GM_EVENT_GPU_PRE_RESET; break;
        case AMDSMI_EVT_NOTIF_GPU_POST_RESET: e.code = GM_EVENT_GPU_POST_RESET; break;
        case AMDSMI_EVT_NOTIF_PROCESS_START: e.code = GM_EVENT_PROCESS_START; break;
        case AMDSMI_EVT_NOTIF_PROCESS_END: e.code = GM_EVENT_PROCESS_END; break;
        default: continue;
      }
      out->push_back(e);
    }
    return GM_ST_OK;
  }

 private:
  const char* errstr(amdsmi_status_t rc) {
    const char* s = nullptr;
    if (api_.amdsmi_status_code_to_string && api_.amdsmi_status_code_to_string(rc, &s) == AMDSMI_STATUS_SUCCESS && s)
      return s;
    return "amdsmi error";
  }

  gm_status_t enumerate() {
    uint32_t nsock = 0;
    amdsmi_status_t rc = api_.amdsmi_get_socket_handles(&nsock, nullptr);
    if (rc != AMDSMI_STATUS_SUCCESS) return GM_ST_INIT_ERROR;
    std::vector<amdsmi_socket_handle> socks(nsock);
    if (nsock && api_.amdsmi_get_socket_handles(&nsock, socks.data()) != AMDSMI_STATUS_SUCCESS) return GM_ST_INIT_ERROR;
    std::vector<Dev> found;
    for (uint32_t s = 0; s < nsock; s++) {
      uint32_t np = 0;
      if (api_.amdsmi_get_processor_handles(socks[s], &np, nullptr) != AMDSMI_STATUS_SUCCESS || !np) continue;
      std::vector<amdsmi_processor_handle> procs(np);
      if (api_.amdsmi_get_processor_handles(socks[s], &np, procs.data()) != AMDSMI_STATUS_SUCCESS) continue;
      for (uint32_t i = 0; i < np; i++) {
        processor_type_t t = AMDSMI_PROCESSOR_TYPE_UNKNOWN;
        if (api_.amdsmi_get_processor_type(procs[i], &t) != AMDSMI_STATUS_SUCCESS || t != AMDSMI_PROCESSOR_TYPE_AMD_GPU)
          continue;
        Dev d;
        d.h = procs[i];
        amdsmi_bdf_t bdf;
        bdf.as_uint = 0;
        if (api_.amdsmi_get_gpu_device_bdf(procs[i], &bdf) == AMDSMI_STATUS_SUCCESS) {
          d.bdf_key = ((uint64_t)bdf.domain_number << 16) | ((uint64_t)bdf.bus_number << 8) |
                      ((uint64_t)bdf.device_number << 3) | bdf.function_number;
          d.st.pci_bdf = strprintf("%04llx:%02x:%02x.%x", (unsigned long long)bdf.domain_number, (unsigned)bdf.bus_number,
                                   (unsigned)bdf.device_number, (unsigned)bdf.function_number);
        }
        if (api_.amdsmi_get_gpu_kfd_info) {
          amdsmi_kfd_info_t k;
          memset(&k, 0, sizeof k);
          if (api_.amdsmi_get_gpu_kfd_info(procs[i], &k) == AMDSMI_STATUS_SUCCESS && valid(k.current_partition_id))
            d.partition_id = k.current_partition_id;
        }
        found.push_back(d);
      }
    }
    // Stable index: PCI address order, then partition.
    std::sort(found.begin(), found.end(), [](const Dev& a, const Dev& b) {
      return a.bdf_key != b.bdf_key ? a.bdf_key < b.bdf_key : a.partition_id < b.partition_id;
    });
    if (found.size() > GM_MAX_GPUS) found.resize(GM_MAX_GPUS);
    prof_tried_ = false;  // the device table is rebuilt: map the counter plugin's agents again
    for (size_t i = 0; i < found.size(); i++) {
      found[i].st.index = (uint32_t)i;
      read_static(found[i]);
    }
    devs_.swap(found);
    close_tables(found);  // the table before this enumeration
    GM_LOGI("amdsmi: %zu GPU(s)", devs_.size());
    return GM_ST_OK;
  }

  void read_static(Dev& d) {
    GpuStatic& s = d.st;
    char buf[AMDSMI_MAX_STRING_LENGTH];
    unsigned int len = sizeof buf;
    if (api_.amdsmi_get_gpu_device_uuid(d.h, &len, buf) == AMDSMI_STATUS_SUCCESS) s.uuid = buf;
    if (api_.amdsmi_get_gpu_asic_info) {
      amdsmi_asic_info_t a;
      memset(&a, 0, sizeof a);
      if (api_.amdsmi_get_gpu_asic_info(d.h, &a) == AMDSMI_STATUS_SUCCESS) {
        s.name = trim(a.market_name);
        // Without libdrm's amdgpu.ids the library only knows the generic family
        // string; the Instinct PCI device ids are few enough to name here.
        if (s.name.empty() || s.name == "AMD Radeon Graphics") {
          static const struct { uint64_t id; const char* name; } kInstinct[] = {
              {0x75a3, "AMD Instinct MI355X"}, {0x75a0, "AMD Instinct MI350X"}, {0x74a5, "AMD Instinct MI325X"},
              {0x74a1, "AMD Instinct MI300X"}, {0x74a9, "AMD Instinct MI300X HF"}, {0x74a0, "AMD Instinct MI300A"},
              {0x740f, "AMD Instinct MI210"},  {0x740c, "AMD Instinct MI250X"},   {0x7408, "AMD Instinct MI250X"}};
          for (auto& k : kInstinct)
            if (k.id == a.device_id) s.name = k.name;
        }
        s.brand = trim(a.vendor_name);
        s.serial = trim(a.asic_serial);
        if (valid(a.num_of_compute_units)) s.num_cu = a.num_of_compute_units;
        if (valid(a.oam_id)) s.oam_id = (int)a.oam_id;
        s.partition_id = (int)d.partition_id;
        if (valid(a.target_graphics_version)) s.gfx_target = strprintf("gfx%llx", (unsigned long long)a.target_graphics_version);
      }
    }
    if (api_.amdsmi_get_gpu_board_info) {
      amdsmi_board_info_t b;
      memset(&b, 0, sizeof b);
      if (api_.amdsmi_get_gpu_board_info(d.h, &b) == AMDSMI_STATUS_SUCCESS) {
        if (*b.product_serial) s.serial = trim(b.product_serial);
        if (s.name.empty() && *b.product_name) s.name = trim(b.product_name);
      }
    }
    if (api_.amdsmi_get_gpu_vbios_info) {
      amdsmi_vbios_info_t v;
      memset(&v, 0, sizeof v);
      if (api_.amdsmi_get_gpu_vbios_info(d.h, &v) == AMDSMI_STATUS_SUCCESS) s.vbios = trim(v.version);
    }
    if (api_.amdsmi_get_gpu_driver_info) {
      amdsmi_driver_info_t v;
      memset(&v, 0, sizeof v);
      if (api_.amdsmi_get_gpu_driver_info(d.h, &v) == AMDSMI_STATUS_SUCCESS) s.driver_version = trim(v.driver_version);
    }
    if (api_.amdsmi_get_fw_info) {
      amdsmi_fw_info_t f;
      memset(&f, 0, sizeof f);
      if (api_.amdsmi_get_fw_info(d.h, &f) == AMDSMI_STATUS_SUCCESS) {
        for (unsigned i = 0; i < f.num_fw_info && i < AMDSMI_FW_ID__MAX; i++) {
          const char* tag = nullptr;
          if (f.fw_info_list[i].fw_id == AMDSMI_FW_ID_PM) tag = "pm";
          else if (f.fw_info_list[i].fw_id == AMDSMI_FW_ID_PSP_SOSDRV) tag = "sos";
          if (tag) s.firmware += strprintf("%s%s=%llx", s.firmware.empty() ? "" : ",", tag,
                                           (unsigned long long)f.fw_info_list[i].fw_version);
        }
      }
    }
    if (api_.amdsmi_get_gpu_enumeration_info) {
      amdsmi_enumeration_info_t e;
      memset(&e, 0, sizeof e);
      if (api_.amdsmi_get_gpu_enumeration_info(d.h, &e) == AMDSMI_STATUS_SUCCESS) {
        if (valid(e.drm_render)) {
          s.drm_render_minor = (int)e.drm_render;
          s.device_path = strprintf("/dev/dri/renderD%u", e.drm_render);
        }
        if (valid(e.hip_id)) s.hip_id = (int)e.hip_id;
      }
    }
    if (api_.amdsmi_get_power_cap_info) {
      amdsmi_power_cap_info_t p;
      memset(&p, 0, sizeof p);
      if (api_.amdsmi_get_power_cap_info(d.h, 0, &p) == AMDSMI_STATUS_SUCCESS) {
        s.power_cap_w = (double)p.power_cap / 1e6;  // µW on bare metal
        // 0 is the library's "not reported" (seen on MI355X with ROCm 7.2), not a 0 W default.
        if (p.default_power_cap > 0) s.power_cap_default_w = (double)p.default_power_cap / 1e6;
        s.power_cap_max_w = (double)p.max_power_cap / 1e6;
      }
    }
    if (api_.amdsmi_get_gpu_memory_total) {
      uint64_t b = 0;
      if (api_.amdsmi_get_gpu_memory_total(d.h, AMDSMI_MEM_TYPE_VRAM, &b) == AMDSMI_STATUS_SUCCESS) s.vram_total_mib = (int64_t)(b >> 20);
      if (api_.amdsmi_get_gpu_memory_total(d.h, AMDSMI_MEM_TYPE_VIS_VRAM, &b) == AMDSMI_STATUS_SUCCESS) s.vis_vram_total_mib = (int64_t)(b >> 20);
    }
    if (api_.amdsmi_get_pcie_info) {
      amdsmi_pcie_info_t p;
      memset(&p, 0, sizeof p);
      if (api_.amdsmi_get_pcie_info(d.h, &p) == AMDSMI_STATUS_SUCCESS) {
        s.pcie_max_width = p.pcie_static.max_pcie_width;
        s.pcie_max_gen = p.pcie_static.max_pcie_interface_version;
        if (!s.pcie_max_gen || !valid(p.pcie_static.max_pcie_interface_version))
          s.pcie_max_gen = (uint32_t)pcie_gen_from_speed_dgts(p.pcie_static.max_pcie_speed / 100);  // MT/s -> 0.1 GT/s
      }
    }
    if (api_.amdsmi_get_clock_info) {
      amdsmi_clk_info_t c;
      memset(&c, 0, sizeof c);
      if (api_.amdsmi_get_clock_info(d.h, AMDSMI_CLK_TYPE_GFX, &c) == AMDSMI_STATUS_SUCCESS && valid(c.max_clk)) s.gfx_clock_max_mhz = c.max_clk;
      if (api_.amdsmi_get_clock_info(d.h, AMDSMI_CLK_TYPE_MEM, &c) == AMDSMI_STATUS_SUCCESS && valid(c.max_clk)) s.mem_clock_max_mhz = c.max_clk;
    }
    if (api_.amdsmi_get_gpu_compute_partition) {
      char p[32] = {0};
      if (api_.amdsmi_get_gpu_compute_partition(d.h, p, sizeof p) == AMDSMI_STATUS_SUCCESS) s.compute_partition = trim(p);
    }
    if (api_.amdsmi_get_gpu_memory_partition) {
      char p[32] = {0};
      if (api_.amdsmi_get_gpu_memory_partition(d.h, p, sizeof p) == AMDSMI_STATUS_SUCCESS) s.memory_partition = trim(p);
    }
    if (api_.amdsmi_topo_get_numa_node_number) {
      uint32_t n = 0;
      if (api_.amdsmi_topo_get_numa_node_number(d.h, &n) == AMDSMI_STATUS_SUCCESS && valid(n)) s.numa_node = (int)n;
    }
    if (!s.pci_bdf.empty()) {
      // Same sysfs sources as the reference (bindings/go/nvml/nvml.go:294-312, bindings/go/dcgm/topology.go:90-96).
      if (s.numa_node < 0) {
        std::string n = sysfs_trim(sysfs_root() + "/bus/pci/devices/" + s.pci_bdf + "/numa_node");
        if (!n.empty() && atoi(n.c_str()) >= 0) s.numa_node = atoi(n.c_str());
      }
      s.cpu_affinity = sysfs_trim(sysfs_root() + "/bus/pci/devices/" + s.pci_bdf + "/local_cpulist");
    }
    if (api_.amdsmi_get_energy_count) {
      uint64_t acc = 0, ts = 0;
      float res = 0;
      if (api_.amdsmi_get_energy_count(d.h, &acc, &res, &ts) == AMDSMI_STATUS_SUCCESS && res > 0) d.energy_resolution_uj = res;
    }
    // XCD count: XCC entries of the metrics table that carry a clock.
    amdsmi_gpu_metrics_t m;
    memset(&m, 0xff, sizeof m);
    if (api_.amdsmi_get_gpu_metrics_info(d.h, &m) == AMDSMI_STATUS_SUCCESS) {
      uint32_t n = 0;
      for (int i = 0; i < AMDSMI_MAX_NUM_GFX_CLKS; i++)
        if (valid(m.current_gfxclks[i]) && m.current_gfxclks[i]) n++;
      s.num_xcd = n;
    }
    // Engine support: CDNA-class (gfx9xx) devices with a metrics table.
    s.supported = !s.uuid.empty() && !s.pci_bdf.empty();
    open_table(d);
  }

  static void close_tables(std::vector<Dev>& v) {
    for (auto& d : v)
      if (d.table_fd >= 0) close(d.table_fd), d.table_fd = -1;
  }

  // One comparison of the direct decoding with the library's, on bytes that did not change under the library's
  // read: 1 = equal, 0 = different (why), -1 = no clean pair this time.
  int compare_with_library(Dev& d, std::string* why) {
    uint8_t r0[4096], r1[4096];
    for (int attempt = 0; attempt < 8; attempt++) {
      const ssize_t n0 = pread(d.table_fd, r0, sizeof r0, 0);
      MetricsTable mine, theirs;
      if (n0 <= 0) { *why = std::string("read: ") + strerror(errno); return 0; }
      if (!metrics_table_decode(r0, (size_t)n0, &mine, why)) return 0;  // a layout this build does not know
      amdsmi_gpu_metrics_t a;
      memset(&a, 0xff, sizeof a);
      if (api_.amdsmi_get_gpu_metrics_info(d.h, &a) != AMDSMI_STATUS_SUCCESS) { *why = "the library's read failed"; return -1; }
      const ssize_t n1 = pread(d.table_fd, r1, sizeof r1, 0);
      if (n0 != n1) { *why = "short read"; return 0; }
      if (memcmp(r0, r1, (size_t)n0)) continue;  // the firmware refreshed the table in between
      from_library(a, &theirs);
      *why = metrics_table_diff(mine, theirs);
      return why->empty() ? 1 : 0;
    }
    *why = "the table kept changing";
    return -1;
  }

  void open_table(Dev& d) {
    const char* mode = getenv("GPUMON_METRICS_TABLE");  // auto (default) | library
    if (mode && !strcmp(mode, "library")) { d.table_note = "switched off by GPUMON_METRICS_TABLE"; return; }
    if (d.st.pci_bdf.empty()) { d.table_note = "no PCI address"; return; }
    const std::string path = sysfs_root() + "/bus/pci/devices/" + d.st.pci_bdf + "/gpu_metrics";
    d.table_fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (d.table_fd < 0) { d.table_note = path + ": " + strerror(errno); return; }
    std::string why;
    const int same = compare_with_library(d, &why);
    if (same != 1) {
      d.table_note = same ? "not checked: " + why : why.rfind("table ", 0) == 0 || why.rfind("read", 0) == 0 ? why : "differs from the library: " + why;
      close(d.table_fd);
      d.table_fd = -1;
      GM_LOGI("amdsmi: gpu %u metrics table through the library (%s)", d.st.index, d.table_note.c_str());
      return;
    }
    d.table_checked_mono = monotonic_us();
    GM_LOGI("amdsmi: gpu %u metrics table read directly from %s", d.st.index, path.c_str());
  }

  // The table for one sample: the direct read where it is on and healthy, else the library.
  bool read_table(Dev& d, MetricsTable* m, amdsmi_status_t* rc) {
    *rc = AMDSMI_STATUS_SUCCESS;
    if (d.table_fd >= 0) {
      const int64_t now = monotonic_us();
      std::string why;
      if (now - d.table_checked_mono > kTableRecheckUs) {
        d.table_checked_mono = now;
        if (compare_with_library(d, &why) == 0) {
          GM_LOGW("amdsmi: gpu %u direct metrics table no longer matches the library (%s): back to the library", d.st.index, why.c_str());
          d.table_note = "differs from the library: " + why;
          close(d.table_fd);
          d.table_fd = -1;
        }
      }
      if (d.table_fd >= 0) {
        uint8_t raw[4096];
        const ssize_t n = pread(d.table_fd, raw, sizeof raw, 0);
        if (n > 0 && metrics_table_decode(raw, (size_t)n, m, &why)) return true;
        GM_LOGW("amdsmi: gpu %u direct metrics table read failed (%s): back to the library", d.st.index, n > 0 ? why.c_str() : strerror(errno));
        d.table_note = n > 0 ? why : std::string("read: ") + strerror(errno);
        close(d.table_fd);
        d.table_fd = -1;
      }
    }
    amdsmi_gpu_metrics_t a;
    memset(&a, 0xff, sizeof a);
    *rc = api_.amdsmi_get_gpu_metrics_info(d.h, &a);
    if (*rc != AMDSMI_STATUS_SUCCESS) return false;
    from_library(a, m);
    return true;
  }

  bool sample_metrics(Dev& d, uint32_t sources, Sample* out) {
    MetricsTable m;
    amdsmi_status_t rc;
    if (!read_table(d, &m, &rc)) {
      out->blank_untouched(sources & (SRC_METRICS | SRC_VIOLATION | SRC_ENERGY | SRC_PROF), blank_reason(rc));
      return false;
    }
    auto set16 = [&](uint16_t id, uint16_t v) { if (valid(v)) out->set_i(id, v); else out->set_blank(id, GM_VS_NOT_SUPPORTED); };
    auto set32 = [&](uint16_t id, uint32_t v) { if (valid(v)) out->set_i(id, v); else out->set_blank(id, GM_VS_NOT_SUPPORTED); };
    auto set64 = [&](uint16_t id, uint64_t v) { if (valid(v)) out->set_i(id, (int64_t)v); else out->set_blank(id, GM_VS_NOT_SUPPORTED); };

    double gr_active = NAN;
    if (sources & (SRC_METRICS | SRC_PROF)) {
      // Per-XCD busy, flattened over partitions.
      int nx = 0;
      double sum = 0;
      for (int p = 0; p < AMDSMI_MAX_NUM_XCP && nx < GM_MAX_XCD; p++)
        for (int x = 0; x < AMDSMI_MAX_NUM_XCC && nx < GM_MAX_XCD; x++) {
          uint32_t b = m.xcp_stats[p].gfx_busy_inst[x];
          if (!valid(b)) continue;
          if (sources & SRC_METRICS) out->set_i(GM_FI_XCD_BUSY_0 + nx, b);
          sum += b;
          nx++;
        }
      if (nx) gr_active = sum / nx;
    }

    if (sources & SRC_METRICS) {
      // gfx clock: mean over the XCDs that report one
      int n = 0;
      uint64_t sum = 0;
      for (int i = 0; i < AMDSMI_MAX_NUM_GFX_CLKS; i++)
        if (valid(m.current_gfxclks[i]) && m.current_gfxclks[i]) { sum += m.current_gfxclks[i]; n++; }
      if (n) out->set_i(GM_FI_GFX_CLOCK, (int64_t)(sum / n)); else set16(GM_FI_GFX_CLOCK, m.current_gfxclk);
      set16(GM_FI_MEM_CLOCK, m.current_uclk);
      set16(GM_FI_SOC_CLOCK, valid(m.current_socclks[0]) ? m.current_socclks[0] : m.current_socclk);
      set16(GM_FI_VCLK, valid(m.current_vclk0s[0]) ? m.current_vclk0s[0] : m.current_vclk0);
      set16(GM_FI_DCLK, valid(m.current_dclk0s[0]) ? m.current_dclk0s[0] : m.current_dclk0);
      set16(GM_FI_GPU_TEMP, m.temperature_hotspot);
      set16(GM_FI_EDGE_TEMP, m.temperature_edge);
      set16(GM_FI_VR_GFX_TEMP, m.temperature_vrgfx);
      set16(GM_FI_VR_SOC_TEMP, m.temperature_vrsoc);
      set16(GM_FI_VR_MEM_TEMP, m.temperature_vrmem);
      uint16_t hbm_max = 0;
      bool any_hbm = false;
      for (int i = 0; i < GM_MAX_HBM_STACKS; i++) {
        if (i < AMDSMI_NUM_HBM_INSTANCES && valid(m.temperature_hbm[i]) && m.temperature_hbm[i]) {
          out->set_i(GM_FI_HBM_STACK_TEMP_0 + i, m.temperature_hbm[i]);
          hbm_max = std::max(hbm_max, m.temperature_hbm[i]);
          any_hbm = true;
        } else {
          out->set_blank(GM_FI_HBM_STACK_TEMP_0 + i, GM_VS_NOT_SUPPORTED);
        }
      }
      if (valid(m.temperature_mem)) out->set_i(GM_FI_MEM_TEMP, m.temperature_mem);
      else if (any_hbm) out->set_i(GM_FI_MEM_TEMP, hbm_max);
      else out->set_blank(GM_FI_MEM_TEMP, GM_VS_NOT_SUPPORTED);
      if (valid(m.current_socket_power)) out->set_d(GM_FI_POWER_USAGE, m.current_socket_power);
      else if (valid(m.average_socket_power)) out->set_d(GM_FI_POWER_USAGE, m.average_socket_power);
      else out->set_blank(GM_FI_POWER_USAGE, GM_VS_NOT_SUPPORTED);
      // The metrics table carries one combined PCIe accumulator, no TX/RX split.  The directional counters come
      // from the driver's one-second throughput measurement where the host offers it (pcie_loop), and stay blank
      // otherwise rather than carry a guess.
      {
        std::lock_guard<std::mutex> pl(pcie_mu_);
        auto it = pcie_acc_.find(d.bdf_key);
        if (it != pcie_acc_.end() && it->second.readings) {
          out->set_i(GM_FI_PCIE_TX_KB, (int64_t)it->second.tx_kb);
          out->set_i(GM_FI_PCIE_RX_KB, (int64_t)it->second.rx_kb);
        } else {
          out->set_blank(GM_FI_PCIE_TX_KB, GM_VS_NOT_SUPPORTED);
          out->set_blank(GM_FI_PCIE_RX_KB, GM_VS_NOT_SUPPORTED);
        }
      }
      set64(GM_FI_PCIE_REPLAY_COUNTER, m.pcie_replay_count_acc);
      if (valid(m.pcie_bandwidth_inst)) out->set_i(GM_FI_PCIE_BW_INST, (int64_t)(m.pcie_bandwidth_inst / 8));  // Mb/s -> MB/s
      else out->set_blank(GM_FI_PCIE_BW_INST, GM_VS_NOT_SUPPORTED);
      set64(GM_FI_PCIE_L0_TO_RECOVERY, m.pcie_l0_to_recov_count_acc);
      set32(GM_FI_PCIE_NAK_SENT, m.pcie_nak_sent_count_acc);
      set32(GM_FI_PCIE_NAK_RCVD, m.pcie_nak_rcvd_count_acc);
      if (valid(m.pcie_link_speed)) out->set_i(GM_FI_DEV_PCIE_GEN, pcie_gen_from_speed_dgts(m.pcie_link_speed));
      else out->set_blank(GM_FI_DEV_PCIE_GEN, GM_VS_NOT_SUPPORTED);
      set16(GM_FI_DEV_PCIE_WIDTH, m.pcie_link_width);
      if (valid(m.average_gfx_activity)) out->set_i(GM_FI_GFX_UTIL, m.average_gfx_activity);
      else if (!std::isnan(gr_active)) out->set_i(GM_FI_GFX_UTIL, (int64_t)llround(gr_active));
      else out->set_blank(GM_FI_GFX_UTIL, GM_VS_NOT_SUPPORTED);
      set16(GM_FI_MEM_UTIL, m.average_umc_activity);
      out->set_blank(GM_FI_ENC_UTIL, GM_VS_NOT_SUPPORTED);  // no encoder on CDNA parts
      if (valid(m.average_mm_activity)) out->set_i(GM_FI_DEC_UTIL, m.average_mm_activity);
      else {
        int nv = 0;
        uint64_t sv = 0;
        for (int i = 0; i < AMDSMI_MAX_NUM_VCN; i++)
          if (valid(m.vcn_activity[i])) { sv += m.vcn_activity[i]; nv++; }
        if (nv) out->set_i(GM_FI_DEC_UTIL, (int64_t)(sv / nv)); else out->set_blank(GM_FI_DEC_UTIL, GM_VS_NOT_SUPPORTED);
      }
      out->set_blank(GM_FI_FAN_SPEED, GM_VS_NOT_SUPPORTED);  // the table reports RPM, and OAM modules have no fan
      int up = 0, down = 0;
      uint64_t bw = 0;
      bool any_bw = false;
      for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) {
        set64(GM_FI_XGMI_LINK_RX_KB_0 + i, m.xgmi_read_data_acc[i]);
        set64(GM_FI_XGMI_LINK_TX_KB_0 + i, m.xgmi_write_data_acc[i]);
        if (valid(m.xgmi_read_data_acc[i])) { bw += m.xgmi_read_data_acc[i]; any_bw = true; }
        if (valid(m.xgmi_write_data_acc[i])) { bw += m.xgmi_write_data_acc[i]; any_bw = true; }
        uint16_t ls = m.xgmi_link_status[i];
        if (!valid(ls) || ls == AMDSMI_XGMI_LINK_DISABLE) {
          out->set_blank(GM_FI_XGMI_LINK_STATUS_0 + i, GM_VS_NOT_SUPPORTED);
        } else {
          out->set_i(GM_FI_XGMI_LINK_STATUS_0 + i, ls == AMDSMI_XGMI_LINK_UP ? 1 : 0);
          if (ls == AMDSMI_XGMI_LINK_UP) up++; else down++;
        }
      }
      if (any_bw) out->set_i(GM_FI_XGMI_BANDWIDTH_TOTAL, (int64_t)bw); else out->set_blank(GM_FI_XGMI_BANDWIDTH_TOTAL, GM_VS_NOT_SUPPORTED);
      set16(GM_FI_XGMI_LINK_WIDTH, m.xgmi_link_width);
      set16(GM_FI_XGMI_LINK_SPEED, m.xgmi_link_speed);
      if (up + down) {
        out->set_i(GM_FI_XGMI_LINKS_UP, up);
        out->set_i(GM_FI_XGMI_LINKS_DOWN, down);
      } else {
        out->set_blank(GM_FI_XGMI_LINKS_UP, GM_VS_NOT_SUPPORTED);
        out->set_blank(GM_FI_XGMI_LINKS_DOWN, GM_VS_NOT_SUPPORTED);
      }
    }

    if (sources & SRC_ENERGY) {
      if (valid(m.energy_accumulator)) out->set_i(GM_FI_TOTAL_ENERGY, (int64_t)((double)m.energy_accumulator * d.energy_resolution_uj / 1000.0));
      else out->set_blank(GM_FI_TOTAL_ENERGY, GM_VS_NOT_SUPPORTED);
    }

    if (sources & SRC_VIOLATION) {
      if (!d.violation_checked) {
        d.violation_checked = true;
        check_violation_source(d);
      }
      // Residency accumulators count firmware accumulation cycles; the cycle
      // length is measured from the firmware timestamp (10 ns units).
      if (valid(m.accumulation_counter) && valid(m.firmware_timestamp) && d.have_prev &&
          m.accumulation_counter > d.prev_acc_counter && m.firmware_timestamp > d.prev_fw_ts) {
        double t = (double)(m.firmware_timestamp - d.prev_fw_ts) * 0.01 / (double)(m.accumulation_counter - d.prev_acc_counter);
        if (t > 1.0 && t < 1e6) d.acc_tick_us = t;
      }
      const uint64_t res[5] = {m.ppt_residency_acc, m.socket_thm_residency_acc, m.hbm_thm_residency_acc,
                               m.vr_thm_residency_acc, m.prochot_residency_acc};
      // Tick counts become durations incrementally: the first reading is priced at the current tick length, every
      // later one adds its own delta at the tick length measured then.  Re-pricing the whole accumulator whenever the
      // estimate of the tick length moves would make a duration run backwards.
      auto us = [&](int slot, uint64_t ticks) {
        Dev::Residency& r = d.residency[slot];
        if (!r.seen || ticks < r.ticks) r.us = (double)ticks * d.acc_tick_us;  // first reading, or the firmware restarted the count
        else r.us += (double)(ticks - r.ticks) * d.acc_tick_us;
        r.seen = true;
        r.ticks = ticks;
        return (int64_t)r.us;
      };
      if (valid(res[0])) out->set_i(GM_FI_POWER_VIOLATION, us(0, res[0])); else out->set_blank(GM_FI_POWER_VIOLATION, GM_VS_NOT_SUPPORTED);
      if (valid(res[1]) || valid(res[2])) out->set_i(GM_FI_THERMAL_VIOLATION, us(1, (valid(res[1]) ? res[1] : 0) + (valid(res[2]) ? res[2] : 0)));
      else out->set_blank(GM_FI_THERMAL_VIOLATION, GM_VS_NOT_SUPPORTED);
      if (valid(res[3]) || valid(res[4])) out->set_i(GM_FI_BOARD_LIMIT_VIOLATION, us(2, (valid(res[3]) ? res[3] : 0) + (valid(res[4]) ? res[4] : 0)));
      else out->set_blank(GM_FI_BOARD_LIMIT_VIOLATION, GM_VS_NOT_SUPPORTED);
      out->set_blank(GM_FI_SYNC_BOOST_VIOLATION, GM_VS_NOT_SUPPORTED);
      // Low-utilisation and below-host-limit residency are kept per XCC.  The contract metric is a duration of the
      // DEVICE ("throttling duration ... (in us)"), which can never advance faster than the wall clock: take the
      // XCC that spent longest in the state, not the sum over XCCs (eight of them would count eight seconds a second).
      uint64_t low_xcc[AMDSMI_MAX_NUM_XCP * AMDSMI_MAX_NUM_XCC], host_xcc[AMDSMI_MAX_NUM_XCP * AMDSMI_MAX_NUM_XCC];
      for (int p = 0; p < AMDSMI_MAX_NUM_XCP; p++)
        for (int x = 0; x < AMDSMI_MAX_NUM_XCC; x++) {
          uint64_t b = m.xcp_stats[p].gfx_below_host_limit_total_acc[x];
          if (!valid(b)) b = m.xcp_stats[p].gfx_below_host_limit_acc[x];
          low_xcc[p * AMDSMI_MAX_NUM_XCC + x] = m.xcp_stats[p].gfx_low_utilization_acc[x];
          host_xcc[p * AMDSMI_MAX_NUM_XCC + x] = b;
        }
      uint64_t low = 0, host = 0;
      const bool any_low = device_residency(low_xcc, AMDSMI_MAX_NUM_XCP * AMDSMI_MAX_NUM_XCC, &low);
      const bool any_host = device_residency(host_xcc, AMDSMI_MAX_NUM_XCP * AMDSMI_MAX_NUM_XCC, &host);
      if (any_low) out->set_i(GM_FI_LOW_UTIL_VIOLATION, us(3, low)); else out->set_blank(GM_FI_LOW_UTIL_VIOLATION, GM_VS_NOT_SUPPORTED);
      // "Below the host limit" is true of every idle GPU and of every power-limited one: it is exported under its
      // own name.  The contract's "throttling due to reliability constraints" has no counterpart in this firmware
      // and stays blank, so that a dashboard that alerts on it does not fire on every idle MI355X.
      if (any_host) out->set_i(GM_FI_GFX_BELOW_HOST_LIMIT, us(4, host)); else out->set_blank(GM_FI_GFX_BELOW_HOST_LIMIT, GM_VS_NOT_SUPPORTED);
      out->set_blank(GM_FI_RELIABILITY_VIOLATION, GM_VS_NOT_SUPPORTED);
      // Active reasons = accumulators that moved between two DISTINCT firmware tables.  The firmware refreshes the
      // table on its own cadence (about once a second); a watch faster than that reads the same table several
      // times, and comparing equal tables would clear every reason between refreshes.
      const bool fresh = !d.have_prev || !valid(m.accumulation_counter) || m.accumulation_counter != d.prev_acc_counter;
      if (fresh) {
        int64_t bits = 0;
        if (d.have_prev) {
          static const int64_t bit_of[5] = {GM_THROTTLE_POWER, GM_THROTTLE_THERMAL, GM_THROTTLE_HBM_THERMAL,
                                            GM_THROTTLE_VR_THERMAL, GM_THROTTLE_PROCHOT};
          for (int i = 0; i < 5; i++)
            if (valid(res[i]) && res[i] > d.prev_res[i]) bits |= bit_of[i];
          if (any_low && low > d.prev_low) bits |= GM_THROTTLE_LOW_UTIL;
          if (any_host && host > d.prev_host) bits |= GM_THROTTLE_HOST_LIMIT;
        }
        d.throttle_bits = bits;
        d.prev_low = any_low ? low : 0;
        d.prev_host = any_host ? host : 0;
        for (int i = 0; i < 5; i++) d.prev_res[i] = valid(res[i]) ? res[i] : 0;
        d.prev_acc_counter = valid(m.accumulation_counter) ? m.accumulation_counter : 0;
        d.prev_fw_ts = valid(m.firmware_timestamp) ? m.firmware_timestamp : 0;
      }
      out->set_i(GM_FI_THROTTLE_STATUS, d.throttle_bits);
      d.have_prev = true;
    }

    if (sources & SRC_PROF) {
      if (!std::isnan(gr_active)) out->set_d(GM_FI_PROF_GR_ENGINE_ACTIVE, gr_active); else out->set_blank(GM_FI_PROF_GR_ENGINE_ACTIVE, GM_VS_NOT_SUPPORTED);
      if (valid(m.average_umc_activity)) out->set_d(GM_FI_PROF_HBM_ACTIVE, m.average_umc_activity); else out->set_blank(GM_FI_PROF_HBM_ACTIVE, GM_VS_NOT_SUPPORTED);
      // CU occupancy from the KFD process table: occupied CUs over all CUs.
      const int64_t now = monotonic_us();
      if (!d.occupancy_read_mono || now - d.occupancy_read_mono >= 1000000) {
        d.occupancy_read_mono = now;
        d.occupancy = std::nan("");
        std::vector<ProcSample> ps;
        if (d.st.num_cu && processes(d.st.index, &ps) == GM_ST_OK) {
          uint64_t cu = 0;
          bool any = false;
          for (auto& p : ps)
            if (valid(p.cu_occupancy)) { cu += p.cu_occupancy; any = true; }
          if (any || ps.empty()) d.occupancy = std::min(100.0, 100.0 * (double)cu / d.st.num_cu);
        }
      }
      if (!std::isnan(d.occupancy)) out->set_d(GM_FI_PROF_CU_OCCUPANCY, d.occupancy);
      else out->set_blank(GM_FI_PROF_CU_OCCUPANCY, GM_VS_NOT_SUPPORTED);
      // The SQ's wave counters only see the waves of the process that programs them, so CU activity is taken from
      // the GRBM block, which is device-wide: the share of cycles the wave launcher of an XCC had work in flight,
      // averaged over the XCCs.  It, the memory-pipeline figure and MFMA pipe activity come from the counter
      // plugin when the operator has switched it on, and are blank otherwise.
      sample_prof_counters(d);
      auto put = [&](uint16_t id, const Dev::ProfValue& pv) {
        // a value is kept for as long as its source takes to come round again (child period or set rotation), plus slack
        if (!std::isnan(pv.v) && now - pv.mono <= prof_keep_us_) out->set_d(id, pv.v);
        else out->set_blank(id, GM_VS_NOT_SUPPORTED);
      };
      put(GM_FI_PROF_MFMA_ACTIVE, d.mfma_active);
      put(GM_FI_PROF_CU_ACTIVE, d.cu_active);
      put(GM_FI_PROF_MEM_PIPE_ACTIVE, d.mem_pipe_active);
      // the firmware's own busy figure wins; the counter fills in where the metrics table has none
      if (std::isnan(gr_active)) put(GM_FI_PROF_GR_ENGINE_ACTIVE, d.gpu_active_ctr);
    }
    return true;
  }

  // Counter plugin: loaded on the first profiling sample if $GPUMON_PROF_COUNTERS is 1, never otherwise (it
  // opens the compute runtime in this process).  Any failure is logged once and leaves the field blank.
  void load_prof_plugin() {
    prof_tried_ = true;
    const char* on = getenv("GPUMON_PROF_COUNTERS");
    if (!on || !*on || !strcmp(on, "0")) return;
    if (!strcmp(on, "child")) {
      // Sampled mode: a short-lived child takes the windows (core/prof/prof_child.h); this process never opens
      // the compute runtime.
      const char* pe = getenv("GPUMON_PROF_PERIOD_S");
      const char* we = getenv("GPUMON_PROF_WINDOW_MS");
      const double period = pe && atof(pe) > 0 ? atof(pe) : 30.0;
      prof_child_.reset(new ProfChildSampler("", period, we && atoi(we) > 0 ? atoi(we) : 250));
      prof_child_->start();
      prof_keep_us_ = (int64_t)(3.0 * std::max(1.0, period) * 1e6) + 2000000;
      GM_LOGI("profiling counters on: one %s --once every %.0f s", ProfChildSampler::default_probe().c_str(), period);
      return;
    }
    // A process that runs HIP work of its own cannot host the plugin: once the plugin has registered and opened
    // HSA, the HIP runtime of the same process finds no device.  Leave the field blank there.
    for (const char* hip : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"})
      if (void* h = dlopen(hip, RTLD_LAZY | RTLD_NOLOAD)) {
        dlclose(h);
        GM_LOGW("profiling counters requested, but this process has the HIP runtime loaded: not loading libgpumon_prof.so");
        return;
      }
    std::vector<std::string> candidates;
    if (const char* e = getenv("GPUMON_PROF_LIB"); e && *e) candidates.push_back(e);
    Dl_info info;
    if (dladdr(reinterpret_cast<void*>(&blank_reason), &info) && info.dli_fname) {
      std::string self = info.dli_fname;
      const size_t slash = self.rfind('/');
      candidates.push_back((slash == std::string::npos ? std::string(".") : self.substr(0, slash)) + "/libgpumon_prof.so");
    }
    candidates.push_back("libgpumon_prof.so");
    for (auto& c : candidates)
      if ((prof_.lib = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
    if (!prof_.lib) {
      GM_LOGW("profiling counters requested but libgpumon_prof.so did not load: %s", dlerror());
      return;
    }
    prof_.open = reinterpret_cast<decltype(prof_.open)>(dlsym(prof_.lib, "gmp_open"));
    prof_.agent_info = reinterpret_cast<decltype(prof_.agent_info)>(dlsym(prof_.lib, "gmp_agent_info"));
    prof_.sample = reinterpret_cast<decltype(prof_.sample)>(dlsym(prof_.lib, "gmp_sample"));
    prof_.last_error = reinterpret_cast<decltype(prof_.last_error)>(dlsym(prof_.lib, "gmp_last_error"));
    prof_.close = reinterpret_cast<decltype(prof_.close)>(dlsym(prof_.lib, "gmp_close"));
    prof_.abi_version = reinterpret_cast<decltype(prof_.abi_version)>(dlsym(prof_.lib, "gmp_abi_version"));
    prof_.set_count = reinterpret_cast<decltype(prof_.set_count)>(dlsym(prof_.lib, "gmp_set_count"));
    if (!prof_.abi_version || !prof_.set_count || prof_.abi_version() != GMP_ABI_VERSION) {
      GM_LOGW("profiling counters unavailable: libgpumon_prof.so has another layout of gmp_sample_t than this library");
      prof_.sample = nullptr;
      return;
    }
    char err[256] = "";
    int n = -1;
    if (!prof_.open || !prof_.agent_info || !prof_.sample || !prof_.last_error || !prof_.close || (n = prof_.open(err, sizeof err)) < 0) {
      GM_LOGW("profiling counters unavailable: %s", err[0] ? err : "plugin has missing entry points");
      prof_.sample = nullptr;
      return;
    }
    // Agent <-> device by PCI address; partitions of one device are agents with the same address, in partition order.
    std::vector<bool> taken((size_t)n, false);
    for (auto& d : devs_)
      for (int a = 0; a < n; a++) {
        gmp_agent_info_t ai;
        if (taken[(size_t)a] || prof_.agent_info(a, &ai) != 0) continue;
        if ((((uint64_t)ai.pci_domain << 16) | ai.pci_location) != d.bdf_key) continue;
        d.prof_agent = a;
        taken[(size_t)a] = true;
        break;
      }
    prof_sets_ = std::max(1, prof_.set_count());
    GM_LOGI("profiling counters on: %d agent(s) through rocprofiler-sdk, %d counter sets in rotation", n, prof_sets_);
  }

  void sample_prof_counters(Dev& d) {
    if (!prof_tried_) load_prof_plugin();
    const int64_t now = monotonic_us();
    auto take = [&](Dev::ProfValue& pv, double v, int64_t at) {
      if (!std::isnan(v)) pv.v = v, pv.mono = at;
    };
    if (prof_child_) {
      ProfWindow w;
      if (!prof_child_->latest(d.st.pci_bdf, &w)) return;  // the values age out by themselves
      take(d.mfma_active, w.mfma_active, w.taken_mono_us);
      take(d.cu_active, w.cu_active, w.taken_mono_us);
      take(d.mem_pipe_active, w.mem_pipe_active, w.taken_mono_us);
      take(d.gpu_active_ctr, w.gpu_active, w.taken_mono_us);
      return;
    }
    if (!prof_.sample || d.prof_agent < 0) return;
    if (d.prof_read_mono && now - d.prof_read_mono < 100000) return;  // one window per 100 ms at most
    // Each window measures one of the rotating signals: a value stays good for two rotations of the sets.
    const int64_t since = d.prof_read_mono ? now - d.prof_read_mono : 1000000;
    prof_keep_us_ = std::max<int64_t>(prof_keep_us_, 2 * prof_sets_ * since + 500000);
    d.prof_read_mono = now;
    gmp_sample_t s;
    const int rc = prof_.sample(d.prof_agent, &s);
    if (rc == 0) {
      take(d.mfma_active, s.mfma_active, now);
      take(d.cu_active, s.cu_active, now);
      take(d.mem_pipe_active, s.mem_pipe_active, now);
      take(d.gpu_active_ctr, s.gpu_active, now);
    } else if (rc < 0) {
      d.mfma_active = d.cu_active = d.mem_pipe_active = d.gpu_active_ctr = Dev::ProfValue();
      if (!prof_warned_) GM_LOGW("profiling counters: %s", prof_.last_error());
      prof_warned_ = true;
    }
  }

  bool sample_vram(Dev& d, Sample* out) {
    if (!api_.amdsmi_get_gpu_memory_usage) {
      out->blank_untouched(SRC_VRAM, GM_VS_NOT_SUPPORTED);
      return false;
    }
    uint64_t used = 0;
    amdsmi_status_t rc = api_.amdsmi_get_gpu_memory_usage(d.h, AMDSMI_MEM_TYPE_VRAM, &used);
    // A loaded driver always holds some VRAM itself, so exactly 0 is not a reading:
    // it is what the library returns where the kernel's mem_info_vram_used counter is
    // not exposed (seen on virtualised hosts).  Report "not supported", not "0 MiB used".
    if (rc == AMDSMI_STATUS_SUCCESS && used == 0 &&
        access((sysfs_root() + "/bus/pci/devices/" + d.st.pci_bdf + "/mem_info_vram_used").c_str(), R_OK) != 0)
      rc = AMDSMI_STATUS_NOT_SUPPORTED;
    if (rc == AMDSMI_STATUS_SUCCESS) {
      int64_t used_mib = (int64_t)(used >> 20);
      out->set_i(GM_FI_VRAM_USED, used_mib);
      if (d.st.vram_total_mib >= 0) out->set_i(GM_FI_VRAM_FREE, d.st.vram_total_mib - used_mib);
    }
    if (d.st.vram_total_mib >= 0) out->set_i(GM_FI_VRAM_TOTAL, d.st.vram_total_mib);  // known even when usage is not
    uint64_t v = 0;
    amdsmi_status_t r2 = api_.amdsmi_get_gpu_memory_usage(d.h, AMDSMI_MEM_TYPE_VIS_VRAM, &v);
    if (r2 == AMDSMI_STATUS_SUCCESS) out->set_i(GM_FI_DEV_VIS_VRAM_USED, (int64_t)(v >> 20));
    r2 = api_.amdsmi_get_gpu_memory_usage(d.h, AMDSMI_MEM_TYPE_GTT, &v);
    if (r2 == AMDSMI_STATUS_SUCCESS) out->set_i(GM_FI_DEV_GTT_USED, (int64_t)(v >> 20));
    out->blank_untouched(SRC_VRAM, blank_reason(rc == AMDSMI_STATUS_SUCCESS ? AMDSMI_STATUS_NOT_SUPPORTED : rc));
    return rc == AMDSMI_STATUS_SUCCESS;
  }

  bool sample_ecc(Dev& d, Sample* out) {
    bool ok = false;
    amdsmi_status_t rc = AMDSMI_STATUS_NOT_SUPPORTED;
    if (api_.amdsmi_get_gpu_total_ecc_count) {
      amdsmi_error_count_t ec;
      memset(&ec, 0, sizeof ec);
      rc = api_.amdsmi_get_gpu_total_ecc_count(d.h, &ec);
      if (rc == AMDSMI_STATUS_SUCCESS) {
        // The driver keeps one set of counters, since it was loaded: that is the volatile pair.  The lifetime totals
        // (RAS EEPROM) are not exposed by the device library, so the aggregate pair is blank rather than a second
        // copy of the same numbers under another name.
        out->set_i(GM_FI_ECC_CE_VOLATILE, (int64_t)ec.correctable_count);
        out->set_i(GM_FI_ECC_UE_VOLATILE, (int64_t)ec.uncorrectable_count);
        out->set_blank(GM_FI_ECC_CE_AGGREGATE, GM_VS_NOT_SUPPORTED);
        out->set_blank(GM_FI_ECC_UE_AGGREGATE, GM_VS_NOT_SUPPORTED);
        out->set_i(GM_FI_ECC_DEFERRED, (int64_t)ec.deferred_count);
        ok = true;
      }
    }
    if (api_.amdsmi_get_gpu_ecc_count) {
      static const amdsmi_gpu_block_t blocks[5] = {AMDSMI_GPU_BLOCK_UMC, AMDSMI_GPU_BLOCK_GFX, AMDSMI_GPU_BLOCK_SDMA,
                                                   AMDSMI_GPU_BLOCK_MMHUB, AMDSMI_GPU_BLOCK_XGMI_WAFL};
      for (int b = 0; b < 5; b++) {
        amdsmi_error_count_t ec;
        memset(&ec, 0, sizeof ec);
        amdsmi_status_t r = api_.amdsmi_get_gpu_ecc_count(d.h, blocks[b], &ec);
        if (r != AMDSMI_STATUS_SUCCESS) {
          out->set_blank(GM_FI_ECC_UMC_CE + 2 * b, blank_reason(r));
          out->set_blank(GM_FI_ECC_UMC_UE + 2 * b, blank_reason(r));
          continue;
        }
        out->set_i(GM_FI_ECC_UMC_CE + 2 * b, (int64_t)ec.correctable_count);
        out->set_i(GM_FI_ECC_UMC_UE + 2 * b, (int64_t)ec.uncorrectable_count);
        if (b == 4) {
          out->set_i(GM_FI_XGMI_CRC_ERRORS, (int64_t)ec.correctable_count);
          out->set_i(GM_FI_XGMI_DATA_ERRORS, (int64_t)ec.uncorrectable_count);
        }
        ok = true;
      }
    }
    // The link layer exposes no replay / recovery COUNTERS for xGMI, only a latched error state; the two
    // contract counters stay blank and the state is its own field.
    out->set_blank(GM_FI_XGMI_REPLAY_ERRORS, GM_VS_NOT_SUPPORTED);
    out->set_blank(GM_FI_XGMI_RECOVERY_ERRORS, GM_VS_NOT_SUPPORTED);
    if (api_.amdsmi_gpu_xgmi_error_status) {
      amdsmi_xgmi_status_t xs = AMDSMI_XGMI_STATUS_NO_ERRORS;
      amdsmi_status_t r = api_.amdsmi_gpu_xgmi_error_status(d.h, &xs);
      if (r == AMDSMI_STATUS_SUCCESS) {
        out->set_i(GM_FI_XGMI_ERROR_STATUS, xs == AMDSMI_XGMI_STATUS_NO_ERRORS ? 0 : xs == AMDSMI_XGMI_STATUS_ERROR ? 1 : 2);
        ok = true;
      } else {
        out->set_blank(GM_FI_XGMI_ERROR_STATUS, blank_reason(r));
      }
    } else {
      out->set_blank(GM_FI_XGMI_ERROR_STATUS, GM_VS_NOT_SUPPORTED);
    }
    out->blank_untouched(SRC_ECC, blank_reason(rc == AMDSMI_STATUS_SUCCESS ? AMDSMI_STATUS_NOT_SUPPORTED : rc));
    return ok;
  }

  bool sample_bad_pages(Dev& d, Sample* out) {
    if (!api_.amdsmi_get_gpu_bad_page_info) {
      out->blank_untouched(SRC_BAD_PAGES, GM_VS_NOT_SUPPORTED);
      return false;
    }
    uint32_t n = 0;
    amdsmi_status_t rc = api_.amdsmi_get_gpu_bad_page_info(d.h, &n, nullptr);
    std::vector<amdsmi_retired_page_record_t> recs;
    if (rc == AMDSMI_STATUS_SUCCESS && n) {
      recs.resize(n);
      rc = api_.amdsmi_get_gpu_bad_page_info(d.h, &n, recs.data());
    }
    if (rc != AMDSMI_STATUS_SUCCESS) {
      out->blank_untouched(SRC_BAD_PAGES, blank_reason(rc));
      return false;
    }
    int64_t reserved = 0, pending = 0;
    for (uint32_t i = 0; i < n && i < recs.size(); i++) {
      if (recs[i].status == AMDSMI_MEM_PAGE_STATUS_PENDING) pending++; else reserved++;
    }
    // The driver retires pages for uncorrectable errors (and correctable storms it escalates); the record does not
    // say which, so every retired page counts as UE and the correctable-cause count is blank, not a made-up 0.
    out->set_blank(GM_FI_BAD_PAGES_CE, GM_VS_NOT_SUPPORTED);
    out->set_i(GM_FI_BAD_PAGES_UE, reserved);
    out->set_i(GM_FI_BAD_PAGES_PENDING, pending);
    return true;
  }

  bool sample_perf(Dev& d, Sample* out) {
    bool ok = false;
    if (api_.amdsmi_get_gpu_perf_level) {
      amdsmi_dev_perf_level_t lvl = AMDSMI_DEV_PERF_LEVEL_UNKNOWN;
      amdsmi_status_t rc = api_.amdsmi_get_gpu_perf_level(d.h, &lvl);
      if (rc == AMDSMI_STATUS_SUCCESS && lvl != AMDSMI_DEV_PERF_LEVEL_UNKNOWN) {
        out->set_i(GM_FI_PERF_LEVEL, (int64_t)lvl);
        ok = true;
      } else {
        out->set_blank(GM_FI_PERF_LEVEL, blank_reason(rc == AMDSMI_STATUS_SUCCESS ? AMDSMI_STATUS_NOT_SUPPORTED : rc));
      }
    }
    if (api_.amdsmi_get_power_cap_info) {
      amdsmi_power_cap_info_t p;
      memset(&p, 0, sizeof p);
      amdsmi_status_t rc = api_.amdsmi_get_power_cap_info(d.h, 0, &p);
      if (rc == AMDSMI_STATUS_SUCCESS) {
        out->set_d(GM_FI_DEV_POWER_CAP, (double)p.power_cap / 1e6);
        ok = true;
      } else {
        out->set_blank(GM_FI_DEV_POWER_CAP, blank_reason(rc));
      }
    }
    out->blank_untouched(SRC_PERF, GM_VS_NOT_SUPPORTED);
    return ok;
  }

  std::string lib_hint_;
  SmiApi api_;
  std::mutex mu_;
  bool inited_ = false, events_tried_ = false, events_any_ = false;
  std::atomic<bool> cper_any_{false};
  int64_t cper_read_mono_ = 0;
  std::mutex pcie_mu_;
  std::map<uint64_t, PcieAcc> pcie_acc_;  // by PCI address: survives a re-enumeration
  std::atomic<bool> pcie_run_{false};
  std::thread pcie_thread_;
  ProfApi prof_;
  bool prof_tried_ = false, prof_warned_ = false;
  int prof_sets_ = 1;
  int64_t prof_keep_us_ = 3000000;
  std::unique_ptr<ProfChildSampler> prof_child_;
  std::vector<Dev> devs_;
};

}  // namespace

std::unique_ptr<Backend> make_amdsmi_backend(const std::string& library_path) {
  return std::unique_ptr<Backend>(new AmdSmiBackend(library_path));
}

}  // namespace gpumon
