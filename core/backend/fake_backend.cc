// FakeBackend: a scripted MI355X node.  Deterministic telemetry (a pure
// function of gpu index and per-GPU sample counter), persistent overrides,
// ramps, per-field "not supported", per-source errors, xGMI link state,
// RAS bumps, process appear/disappear and driver-event injection.  It is the
// interface seam the reference never had (its tests need real hardware:
// bindings/go/nvml/nvml_test.go:18-31, bindings/go/dcgm/dcgm_test.go:18-31).
#include <algorithm>
#include <condition_variable>
#include <deque>
#include <map>
#include <mutex>

#include "../common/util.h"
#include "../prof/prof_child.h"
#include "backend.h"

namespace gpumon {
namespace {

struct Ramp {
  double from, to;
  int steps, done;
};

struct FakeGpu {
  GpuStatic st;
  uint64_t n = 0;  // samples taken
  double load = 0; // 0..100, drives util / power / clocks / temperature
  std::map<uint16_t, double> overrides;
  std::map<uint16_t, Ramp> ramps;
  std::map<uint16_t, uint8_t> blanked;  // field -> status
  uint32_t failing_sources = 0;
  bool link_up[GM_MAX_XGMI_LINKS];
  int64_t ecc_ce[5] = {0, 0, 0, 0, 0}, ecc_ue[5] = {0, 0, 0, 0, 0}, ecc_deferred = 0;
  int64_t bad_ce = 0, bad_ue = 0, bad_pending = 0;
  int64_t energy_mj = 0, pcie_tx_kb = 0, pcie_rx_kb = 0, pcie_replay = 0;
  int64_t xgmi_rx[GM_MAX_XGMI_LINKS], xgmi_tx[GM_MAX_XGMI_LINKS];
  int64_t viol_power = 0, viol_thermal = 0, viol_board = 0, viol_rel = 0;
  uint64_t viol_low_xcc[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // per-XCC low-utilisation residency (us): every idle XCC accrues wall time
  int64_t viol_seen_mono = 0;
  int64_t prev_viol[5] = {0, 0, 0, 0, 0};  // power, thermal, board, low, reliability at the previous sample
  int xgmi_error_state = 0;
  int64_t vram_used_mib = 283;
  std::vector<ProcSample> procs;
};

class FakeBackend : public Backend {
 public:
  explicit FakeBackend(const std::string& spec) {
    for (auto& kv : split(spec, ',')) {
      auto p = kv.find('=');
      if (p == std::string::npos) continue;
      std::string k = trim(kv.substr(0, p)), v = trim(kv.substr(p + 1));
      if (k == "gpus") ngpus_ = (uint32_t)std::max(0, atoi(v.c_str()));
      else if (k == "seed") seed_ = (uint32_t)strtoul(v.c_str(), nullptr, 10);
      else if (k == "fail_init") fail_init_ = atoi(v.c_str()) != 0;
      else if (k == "events") events_supported_ = atoi(v.c_str()) != 0;
      else if (k == "load") default_load_ = atof(v.c_str());
      else if (k == "unsupported_gpu") unsupported_gpu_ = atoi(v.c_str());
      else if (k == "sample_cost_us") sample_cost_us_ = atoi(v.c_str());
      else if (k == "steady") steady_ = atoi(v.c_str()) != 0;  // telemetry independent of the sample count
      else if (k == "partition_mode") partition_mode_ = v;       // e.g. CPX: every entry is one of 8 partitions of a GPU
      // control commands run at the end of init, '+' for the blanks and ';' between commands (a spec has no blanks and
      // its commas separate keys): at_start=bad_pages+gpu=2+pending=3;static+gpu=5+vbios=113-OLD
      else if (k == "at_start") at_start_ = v;
    }
    if (ngpus_ > GM_MAX_GPUS) ngpus_ = GM_MAX_GPUS;
  }

  const char* name() const override { return "fake"; }

  gm_status_t init() override {
    if (fail_init_) return GM_ST_INIT_ERROR;
    gm_status_t st = init_devices();
    if (st != GM_ST_OK) return st;
    std::string cmds = at_start_;
    std::replace(cmds.begin(), cmds.end(), '+', ' ');
    for (auto& c : split(cmds, ';'))
      if (!trim(c).empty() && (st = control(trim(c))) != GM_ST_OK) return st;
    return GM_ST_OK;
  }

  gm_status_t init_devices() {
    std::lock_guard<std::mutex> l(mu_);
    gpus_.assign(ngpus_, FakeGpu());
    for (uint32_t i = 0; i < ngpus_; i++) {
      FakeGpu& g = gpus_[i];
      g.load = default_load_;
      GpuStatic& s = g.st;
      s.index = i;
      s.supported = (int)i != unsupported_gpu_;
      s.uuid = strprintf("GPU-%08x-fa4e-0355-0000-%012x", seed_, i);
      s.name = "AMD Instinct MI355X";
      s.brand = "Instinct";
      s.serial = strprintf("FAKE%08u", 35500000u + i);
      s.vbios = "113-M3550101-100";
      s.firmware = "psp-00.37.01.0b/smu-86.48.0";
      s.driver_version = "6.16.6";
      s.pci_bdf = strprintf("0000:%02x:00.0", 0x05 + 0x10 * i);
      s.gfx_target = "gfx950";
      s.device_path = strprintf("/dev/dri/renderD%d", 128 + (int)i);
      s.cpu_affinity = i < (ngpus_ + 1) / 2 ? "0-63,128-191" : "64-127,192-255";
      s.compute_partition = partition_mode_.empty() ? "SPX" : partition_mode_;
      s.memory_partition = "NPS1";
      if (!partition_mode_.empty() && partition_mode_ != "SPX") {
        const uint32_t per_gpu = partition_mode_ == "DPX" ? 2 : partition_mode_ == "QPX" ? 4 : 8;
        s.partition_id = (int)(i % per_gpu);
      }
      s.numa_node = i < (ngpus_ + 1) / 2 ? 0 : 1;
      s.hip_id = (int)i;
      s.drm_render_minor = 128 + (int)i;
      s.oam_id = (int)i;
      s.num_xcd = 8;
      s.num_cu = 256;
      s.pcie_max_gen = 5;
      s.pcie_max_width = 16;
      s.power_cap_w = 1400;
      s.power_cap_default_w = 1400;
      s.power_cap_max_w = 1400;
      s.vram_total_mib = 294896;  // 288 GB HBM3E as the driver reports it
      s.vis_vram_total_mib = 294896;
      s.gfx_clock_max_mhz = 2400;
      s.mem_clock_max_mhz = 2000;
      for (int l = 0; l < GM_MAX_XGMI_LINKS; l++) {
        g.link_up[l] = l < (int)ngpus_ - 1 && l < 7;
        g.xgmi_rx[l] = g.xgmi_tx[l] = 0;
      }
    }
    return GM_ST_OK;
  }

  void shutdown() override {
    std::lock_guard<std::mutex> l(mu_);
    gpus_.clear();
    ev_cv_.notify_all();
  }

  uint32_t device_count() override {
    std::lock_guard<std::mutex> l(mu_);
    return (uint32_t)gpus_.size();
  }

  gm_status_t static_info(uint32_t gpu, GpuStatic* out) override {
    std::lock_guard<std::mutex> l(mu_);
    if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
    *out = gpus_[gpu].st;
    return GM_ST_OK;
  }

  gm_status_t sample(uint32_t gpu, uint32_t sources, Sample* out) override {
    if (sample_cost_us_ > 0) {
      struct timespec ts = {0, sample_cost_us_ * 1000L};
      nanosleep(&ts, nullptr);
    }
    std::lock_guard<std::mutex> l(mu_);
    if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
    FakeGpu& g = gpus_[gpu];
    uint32_t failing = g.failing_sources & sources;
    uint32_t ok = sources & ~failing;
    g.n++;
    const double load = g.load;
    const int64_t n = (int64_t)g.n;
    const int64_t wobble = steady_ ? 0 : (int64_t)((n + gpu) % 3);
    const int64_t adv = steady_ ? (g.n == 1 ? 1 : 0) : 1;  // counters advance once, then hold, in steady mode

    if (ok & SRC_METRICS) {
      int64_t util = (int64_t)load;
      out->set_i(GM_FI_GFX_CLOCK, load > 0 ? 2400 - (int64_t)(load * 3) : 132);
      out->set_i(GM_FI_MEM_CLOCK, load > 0 ? 2000 : 900);
      out->set_i(GM_FI_SOC_CLOCK, 1143);
      out->set_i(GM_FI_VCLK, 29);
      out->set_i(GM_FI_DCLK, 22);
      int64_t hot = 38 + (int64_t)gpu + (int64_t)(load * 0.5) + wobble;
      out->set_i(GM_FI_GPU_TEMP, hot);
      out->set_i(GM_FI_EDGE_TEMP, hot - 6);
      out->set_i(GM_FI_MEM_TEMP, hot - 3);
      out->set_i(GM_FI_VR_GFX_TEMP, hot - 8);
      out->set_i(GM_FI_VR_SOC_TEMP, hot - 9);
      out->set_i(GM_FI_VR_MEM_TEMP, hot - 10);
      for (int i = 0; i < GM_MAX_HBM_STACKS; i++) out->set_i(GM_FI_HBM_STACK_TEMP_0 + i, hot - 3 - (i & 1));
      double power = 140.0 + gpu + load * 12.0 + (double)wobble;
      out->set_d(GM_FI_POWER_USAGE, power);
      g.pcie_tx_kb += adv * (1200 + (int64_t)(load * 4000));
      g.pcie_rx_kb += adv * (900 + (int64_t)(load * 6000));
      out->set_i(GM_FI_PCIE_TX_KB, g.pcie_tx_kb);
      out->set_i(GM_FI_PCIE_RX_KB, g.pcie_rx_kb);
      out->set_i(GM_FI_PCIE_REPLAY_COUNTER, g.pcie_replay);
      out->set_i(GM_FI_PCIE_BW_INST, 2 + (int64_t)(load * 10));
      out->set_i(GM_FI_PCIE_L0_TO_RECOVERY, 0);
      out->set_i(GM_FI_PCIE_NAK_SENT, 0);
      out->set_i(GM_FI_PCIE_NAK_RCVD, 0);
      out->set_i(GM_FI_DEV_PCIE_GEN, 5);
      out->set_i(GM_FI_DEV_PCIE_WIDTH, 16);
      out->set_i(GM_FI_GFX_UTIL, util);
      out->set_i(GM_FI_MEM_UTIL, (int64_t)(load * 0.6));
      out->set_blank(GM_FI_ENC_UTIL, GM_VS_NOT_SUPPORTED);  // MI355X has no encoder
      out->set_i(GM_FI_DEC_UTIL, 0);
      out->set_blank(GM_FI_FAN_SPEED, GM_VS_NOT_SUPPORTED);  // OAM module: no fan
      int up = 0, down = 0;
      int64_t bw = 0;
      for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) {
        bool wired = i < (int)gpus_.size() - 1 && i < 7;
        if (!wired) {
          out->set_blank(GM_FI_XGMI_LINK_RX_KB_0 + i, GM_VS_NOT_SUPPORTED);
          out->set_blank(GM_FI_XGMI_LINK_TX_KB_0 + i, GM_VS_NOT_SUPPORTED);
          out->set_blank(GM_FI_XGMI_LINK_STATUS_0 + i, GM_VS_NOT_SUPPORTED);
          continue;
        }
        if (g.link_up[i]) {
          g.xgmi_rx[i] += adv * ((int64_t)(load * 150000) + 10);
          g.xgmi_tx[i] += adv * ((int64_t)(load * 150000) + 12);
          up++;
        } else {
          down++;
        }
        bw += g.xgmi_rx[i] + g.xgmi_tx[i];
        out->set_i(GM_FI_XGMI_LINK_RX_KB_0 + i, g.xgmi_rx[i]);
        out->set_i(GM_FI_XGMI_LINK_TX_KB_0 + i, g.xgmi_tx[i]);
        out->set_i(GM_FI_XGMI_LINK_STATUS_0 + i, g.link_up[i] ? 1 : 0);
      }
      out->set_i(GM_FI_XGMI_BANDWIDTH_TOTAL, bw);
      out->set_i(GM_FI_XGMI_LINK_WIDTH, 16);
      out->set_i(GM_FI_XGMI_LINK_SPEED, 38);
      out->set_i(GM_FI_XGMI_LINKS_UP, up);
      out->set_i(GM_FI_XGMI_LINKS_DOWN, down);
      for (int i = 0; i < GM_MAX_XCD; i++) out->set_i(GM_FI_XCD_BUSY_0 + i, util);
    }
    if (ok & SRC_VRAM) {
      int64_t used = g.vram_used_mib;
      for (auto& p : g.procs) used += (int64_t)(p.vram_bytes >> 20);
      out->set_i(GM_FI_VRAM_TOTAL, g.st.vram_total_mib);
      out->set_i(GM_FI_VRAM_USED, used);
      out->set_i(GM_FI_VRAM_FREE, g.st.vram_total_mib - used);
      out->set_i(GM_FI_DEV_VIS_VRAM_USED, used);
      out->set_i(GM_FI_DEV_GTT_USED, 14);
    }
    if (ok & SRC_ECC) {
      int64_t ce = 0, ue = 0;
      for (int b = 0; b < 5; b++) {
        ce += g.ecc_ce[b];
        ue += g.ecc_ue[b];
        out->set_i(GM_FI_ECC_UMC_CE + 2 * b, g.ecc_ce[b]);
        out->set_i(GM_FI_ECC_UMC_UE + 2 * b, g.ecc_ue[b]);
      }
      out->set_i(GM_FI_ECC_CE_VOLATILE, ce);
      out->set_i(GM_FI_ECC_UE_VOLATILE, ue);
      out->set_i(GM_FI_ECC_CE_AGGREGATE, ce);
      out->set_i(GM_FI_ECC_UE_AGGREGATE, ue);
      out->set_i(GM_FI_ECC_DEFERRED, g.ecc_deferred);
      out->set_i(GM_FI_XGMI_CRC_ERRORS, g.ecc_ce[4]);
      out->set_i(GM_FI_XGMI_DATA_ERRORS, g.ecc_ue[4]);
      out->set_blank(GM_FI_XGMI_REPLAY_ERRORS, GM_VS_NOT_SUPPORTED);
      out->set_blank(GM_FI_XGMI_RECOVERY_ERRORS, GM_VS_NOT_SUPPORTED);
      out->set_i(GM_FI_XGMI_ERROR_STATUS, g.xgmi_error_state);
    }
    if (ok & SRC_BAD_PAGES) {
      out->set_i(GM_FI_BAD_PAGES_CE, g.bad_ce);
      out->set_i(GM_FI_BAD_PAGES_UE, g.bad_ue);
      out->set_i(GM_FI_BAD_PAGES_PENDING, g.bad_pending);
    }
    if (ok & SRC_ENERGY) {
      g.energy_mj += adv * (int64_t)((140.0 + gpu + load * 12.0) * 1000.0);
      out->set_i(GM_FI_TOTAL_ENERGY, g.energy_mj);
    }
    if (ok & SRC_PERF) {
      out->set_i(GM_FI_PERF_LEVEL, load > 0 ? 0 : 1);
      out->set_d(GM_FI_DEV_POWER_CAP, g.st.power_cap_w);
    }
    if (ok & SRC_VIOLATION) {
      out->set_i(GM_FI_POWER_VIOLATION, g.viol_power);
      out->set_i(GM_FI_THERMAL_VIOLATION, g.viol_thermal);
      out->set_blank(GM_FI_SYNC_BOOST_VIOLATION, GM_VS_NOT_SUPPORTED);
      out->set_i(GM_FI_BOARD_LIMIT_VIOLATION, g.viol_board);
      // An idle GPU sits in the low-utilisation state on all eight XCCs at once; the device duration is the worst
      // XCC's, which is what keeps it under the wall clock.
      const int64_t mono = monotonic_us();
      if (g.viol_seen_mono && load < 5 && !steady_)
        for (int x = 0; x < 8; x++) g.viol_low_xcc[x] += (uint64_t)((double)(mono - g.viol_seen_mono) * (1.0 - 0.01 * x));
      g.viol_seen_mono = mono;
      uint64_t low = 0;
      device_residency(g.viol_low_xcc, 8, &low);
      out->set_i(GM_FI_LOW_UTIL_VIOLATION, (int64_t)low);
      out->set_i(GM_FI_RELIABILITY_VIOLATION, g.viol_rel);
      const int64_t cur[5] = {g.viol_power, g.viol_thermal, g.viol_board, (int64_t)low, g.viol_rel};
      static const int64_t bit_of[5] = {GM_THROTTLE_POWER, GM_THROTTLE_THERMAL, GM_THROTTLE_PROCHOT, GM_THROTTLE_LOW_UTIL, GM_THROTTLE_HOST_LIMIT};
      int64_t bits = 0;
      for (int i = 0; i < 5; i++) {
        if (g.n > 1 && cur[i] > g.prev_viol[i]) bits |= bit_of[i];
        g.prev_viol[i] = cur[i];
      }
      out->set_i(GM_FI_THROTTLE_STATUS, bits);
    }
    if (ok & SRC_PROF) {
      out->set_d(GM_FI_PROF_GR_ENGINE_ACTIVE, load);
      out->set_d(GM_FI_PROF_CU_ACTIVE, load * 0.9);
      out->set_d(GM_FI_PROF_CU_OCCUPANCY, load * 0.4);
      out->set_d(GM_FI_PROF_MFMA_ACTIVE, load * 0.7);
      out->set_d(GM_FI_PROF_HBM_ACTIVE, load * 0.6);
      out->set_d(GM_FI_PROF_MEM_PIPE_ACTIVE, load * 0.5);
      // $GPUMON_PROF_COUNTERS=child: the counter-derived values come from the short-lived probe child, exactly as
      // with the real device library (the probe is whatever $GPUMON_PROF_PROBE names); blank until it has reported.
      if (const char* e = getenv("GPUMON_PROF_COUNTERS"); e && !strcmp(e, "child")) {
        if (!prof_child_) {
          const char* pe = getenv("GPUMON_PROF_PERIOD_S");
          const char* we = getenv("GPUMON_PROF_WINDOW_MS");
          prof_child_.reset(new ProfChildSampler("", pe && atof(pe) > 0 ? atof(pe) : 30.0, we && atoi(we) > 0 ? atoi(we) : 250));
          prof_child_->start();
        }
        ProfWindow w;
        const bool have = prof_child_->latest(g.st.pci_bdf, &w);
        auto put = [&](uint16_t id, double v) {
          if (have && !std::isnan(v)) out->set_d(id, v);
          else out->set_blank(id, GM_VS_NOT_SUPPORTED);
        };
        put(GM_FI_PROF_MFMA_ACTIVE, w.mfma_active);
        put(GM_FI_PROF_CU_ACTIVE, w.cu_active);
        put(GM_FI_PROF_MEM_PIPE_ACTIVE, w.mem_pipe_active);
      }
    }
    if (failing) out->blank_untouched(failing, GM_VS_ERROR);

    // Scripted modifications win over the model.
    for (auto it = g.ramps.begin(); it != g.ramps.end();) {
      Ramp& r = it->second;
      double x = r.steps <= 1 ? r.to : r.from + (r.to - r.from) * (double)r.done / (double)(r.steps - 1);
      const FieldDef* f = field_def(it->first);
      if (f && (f->source & sources)) {
        out->set_num(it->first, x);
        if (r.done < r.steps - 1) r.done++;
      }
      ++it;
    }
    for (auto& kv : g.overrides) {
      const FieldDef* f = field_def(kv.first);
      if (f && (f->source & sources)) out->set_num(kv.first, kv.second);
    }
    for (auto& kv : g.blanked) {
      const FieldDef* f = field_def(kv.first);
      if (f && (f->source & sources)) out->set_blank(kv.first, kv.second);
    }
    return failing && !ok ? GM_ST_GENERIC_ERROR : GM_ST_OK;
  }

  const char* process_source(uint32_t) override { return "fake"; }
  gm_status_t processes(uint32_t gpu, std::vector<ProcSample>* out) override {
    std::lock_guard<std::mutex> l(mu_);
    if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
    if (gpus_[gpu].failing_sources & SRC_PROCS) return GM_ST_GENERIC_ERROR;
    // Engine time advances while the process is listed.
    for (auto& p : gpus_[gpu].procs) p.gfx_engine_ns += 400000000ull;
    *out = gpus_[gpu].procs;
    return GM_ST_OK;
  }

  gm_status_t peer_link(uint32_t gpu, uint32_t peer, PeerLink* out) override {
    std::lock_guard<std::mutex> l(mu_);
    if (gpu >= gpus_.size() || peer >= gpus_.size() || gpu == peer) return GM_ST_BADPARAM;
    if (topology_unsupported_) return GM_ST_NOT_SUPPORTED;
    PeerLink p;
    if (gpus_.size() <= 8) {
      p.link = GM_LINK_XGMI;
      p.hops = 1;
      p.num_links = 1;
      p.weight = 15;
      p.min_bw_mbs = 50000;
      p.max_bw_mbs = 153600;
    } else {
      // Two xGMI hives bridged over PCIe across the CPU sockets.
      bool same_hive = (gpu / 8) == (peer / 8);
      p.link = same_hive ? GM_LINK_XGMI : GM_LINK_CROSS_CPU;
      p.hops = same_hive ? 1 : 3;
      p.num_links = same_hive ? 1 : 0;
      p.weight = same_hive ? 15 : 72;
      p.min_bw_mbs = same_hive ? 50000 : -1;
      p.max_bw_mbs = same_hive ? 153600 : -1;
    }
    p.p2p_accessible = true;
    *out = p;
    return GM_ST_OK;
  }

  // Fully connected hive of up to eight: link l of GPU g goes to the l-th other GPU in index order.
  static uint32_t peer_of(uint32_t gpu, uint32_t link) { return link < gpu ? link : link + 1; }

  gm_status_t xgmi_links(uint32_t gpu, std::vector<XgmiLink>* out) override {
    std::lock_guard<std::mutex> l(mu_);
    out->clear();
    if (gpu >= gpus_.size()) return GM_ST_BADPARAM;
    if (topology_unsupported_) return GM_ST_NOT_SUPPORTED;
    const uint32_t hive0 = gpu / 8 * 8, hive_n = (uint32_t)std::min<size_t>(8, gpus_.size() - hive0);
    for (uint32_t i = 0; i + 1 < hive_n && i < 7; i++) {
      XgmiLink k;
      k.index = i;
      k.state = gpus_[gpu].link_up[i] ? 1 : 0;
      k.peer_bdf = gpus_[hive0 + peer_of(gpu - hive0, i)].st.pci_bdf;
      k.bit_rate_gbps = 38;
      k.max_bandwidth_gbps = 608;
      k.rx_kb = gpus_[gpu].xgmi_rx[i];
      k.tx_kb = gpus_[gpu].xgmi_tx[i];
      out->push_back(k);
    }
    return GM_ST_OK;
  }

  gm_status_t poll_events(int timeout_ms, std::vector<DeviceEvent>* out) override {
    if (!events_supported_) return GM_ST_NOT_SUPPORTED;
    std::unique_lock<std::mutex> l(mu_);
    cv_wait_for(ev_cv_, l, std::chrono::milliseconds(timeout_ms), [&] { return !events_.empty() || gpus_.empty(); });
    while (!events_.empty()) {
      out->push_back(events_.front());
      events_.pop_front();
    }
    return GM_ST_OK;
  }

  gm_status_t runtime_version(uint32_t* major, uint32_t* minor, std::string* text) override {
    *major = 7, *minor = 2, *text = "7.2.0";
    return GM_ST_OK;
  }

  // "<verb> key=value key=value ..." (keys may also be separated by commas).
  gm_status_t control(const std::string& command) override {
    std::string c = command;
    std::replace(c.begin(), c.end(), ',', ' ');
    std::vector<std::string> tok;
    for (auto& t : split(c, ' '))
      if (!trim(t).empty()) tok.push_back(trim(t));
    if (tok.empty()) return GM_ST_BADPARAM;
    std::string verb = tok[0];
    std::map<std::string, std::string> kv;
    size_t first = 1;
    if (verb.find('=') != std::string::npos) {  // verb omitted: "gpu=1 field=150 value=97"
      verb = "set";
      first = 0;
    }
    for (size_t i = first; i < tok.size(); i++) {
      auto p = tok[i].find('=');
      if (p == std::string::npos) kv[tok[i]] = "1"; else kv[tok[i].substr(0, p)] = tok[i].substr(p + 1);
    }
    auto geti = [&](const char* k, long def) { return kv.count(k) ? strtol(kv[k].c_str(), nullptr, 0) : def; };
    auto getd = [&](const char* k, double def) { return kv.count(k) ? atof(kv[k].c_str()) : def; };
    auto field_of = [&]() -> int {
      if (!kv.count("field")) return -1;
      const std::string& f = kv["field"];
      if (isdigit((unsigned char)f[0])) return atoi(f.c_str());
      const FieldDef* d = field_by_tag(f);
      return d ? d->id : -1;
    };

    std::lock_guard<std::mutex> l(mu_);
    if (verb == "topology_unsupported") {
      topology_unsupported_ = geti("on", 1) != 0;
      return GM_ST_OK;
    }
    std::vector<uint32_t> targets;
    if (!kv.count("gpu") || kv["gpu"] == "*") {
      for (uint32_t i = 0; i < gpus_.size(); i++) targets.push_back(i);
    } else {
      long g = geti("gpu", -1);
      if (g < 0 || (size_t)g >= gpus_.size()) return GM_ST_BADPARAM;
      targets.push_back((uint32_t)g);
    }
    for (uint32_t gi : targets) {
      FakeGpu& g = gpus_[gi];
      if (verb == "set") {
        int f = field_of();
        if (f < 0 || !field_def((uint16_t)f) || !kv.count("value")) return GM_ST_BADPARAM;
        g.overrides[(uint16_t)f] = getd("value", 0);
        g.blanked.erase((uint16_t)f);
      } else if (verb == "clear") {
        int f = field_of();
        if (f < 0) {
          g.overrides.clear(); g.ramps.clear(); g.blanked.clear(); g.failing_sources = 0;
        } else {
          g.overrides.erase((uint16_t)f); g.ramps.erase((uint16_t)f); g.blanked.erase((uint16_t)f);
        }
      } else if (verb == "unsupported" || verb == "noperm") {
        int f = field_of();
        if (f < 0) return GM_ST_BADPARAM;
        g.blanked[(uint16_t)f] = verb == "noperm" ? GM_VS_NO_PERMISSION : GM_VS_NOT_SUPPORTED;
      } else if (verb == "ramp") {
        int f = field_of();
        if (f < 0) return GM_ST_BADPARAM;
        g.ramps[(uint16_t)f] = Ramp{getd("from", 0), getd("to", 0), (int)geti("steps", 2), 0};
      } else if (verb == "fail") {
        g.failing_sources = (uint32_t)geti("sources", 0xffffffff);
      } else if (verb == "load") {
        g.load = getd("util", getd("value", 0));
      } else if (verb == "ecc") {
        int b = (int)geti("block", 0);
        if (b < 0 || b > 4) return GM_ST_BADPARAM;
        g.ecc_ce[b] += geti("ce", 0);
        g.ecc_ue[b] += geti("ue", 0);
        g.ecc_deferred += geti("deferred", 0);
      } else if (verb == "static") {  // what a device reports about itself: one GPU unlike its neighbours
        if (kv.count("vbios")) g.st.vbios = kv["vbios"];
        if (kv.count("firmware")) g.st.firmware = kv["firmware"];
        if (kv.count("power_cap")) g.st.power_cap_w = getd("power_cap", g.st.power_cap_w);
        if (kv.count("compute_partition")) g.st.compute_partition = kv["compute_partition"];
        if (kv.count("supported")) g.st.supported = geti("supported", 1) != 0;
      } else if (verb == "bad_pages") {
        g.bad_ce += geti("ce", 0); g.bad_ue += geti("ue", 0); g.bad_pending += geti("pending", 0);
      } else if (verb == "pcie_replay") {
        g.pcie_replay += geti("count", 1);
      } else if (verb == "violation") {
        g.viol_power += geti("power", 0); g.viol_thermal += geti("thermal", 0);
        g.viol_board += geti("board", 0);
        for (int x = 0; x < 8; x++) g.viol_low_xcc[x] += (uint64_t)geti("low_util", 0);
        g.viol_rel += geti("reliability", 0);
      } else if (verb == "xgmi_error") {
        g.xgmi_error_state = (int)geti("state", 1);
      } else if (verb == "link_down" || verb == "link_up") {
        long lk = geti("link", -1);
        if (lk < 0 || lk >= GM_MAX_XGMI_LINKS) return GM_ST_BADPARAM;
        g.link_up[lk] = verb == "link_up";
      } else if (verb == "proc_add") {
        ProcSample p;
        p.pid = (uint32_t)geti("pid", 0);
        p.name = kv.count("name") ? kv["name"] : "fake_proc";
        p.vram_bytes = (uint64_t)strtoull(kv.count("vram") ? kv["vram"].c_str() : "1073741824", nullptr, 0);
        p.cu_occupancy = (uint32_t)geti("cu", 64);
        g.procs.push_back(p);
      } else if (verb == "proc_del") {
        uint32_t pid = (uint32_t)geti("pid", 0);
        g.procs.erase(std::remove_if(g.procs.begin(), g.procs.end(), [&](const ProcSample& p) { return p.pid == pid; }),
                      g.procs.end());
      } else if (verb == "event") {
        DeviceEvent e;
        e.gpu = gi;
        e.code = (uint32_t)geti("code", GM_EVENT_GPU_PRE_RESET);
        e.ts_us = now_us();
        e.message = kv.count("msg") ? kv["msg"] : "injected";
        events_.push_back(e);
        ev_cv_.notify_all();
      } else {
        return GM_ST_BADPARAM;
      }
    }
    return GM_ST_OK;
  }

 private:
  uint32_t ngpus_ = 8, seed_ = 1;
  std::unique_ptr<ProfChildSampler> prof_child_;
  bool fail_init_ = false, events_supported_ = true, topology_unsupported_ = false, steady_ = false;
  std::string partition_mode_;
  std::string at_start_;
  double default_load_ = 0;
  int unsupported_gpu_ = -1;
  int sample_cost_us_ = 0;
  std::mutex mu_;
  std::condition_variable ev_cv_;
  std::vector<FakeGpu> gpus_;
  std::deque<DeviceEvent> events_;
};

}  // namespace

std::unique_ptr<Backend> make_fake_backend(const std::string& spec) {
  return std::unique_ptr<Backend>(new FakeBackend(spec));
}

// "replicate=N" anywhere in a comma-separated spec wraps the backend so that every GPU appears N times
// (replicate_backend.cc).  For amdsmi the rest of the spec is "lib=PATH" or a bare library path.
std::unique_ptr<Backend> make_backend(const std::string& name, const std::string& spec) {
  uint32_t replicate = 0;
  std::string rest, lib;
  for (auto& kv : split(spec, ',')) {
    const std::string t = trim(kv);
    if (t.compare(0, 10, "replicate=") == 0) replicate = (uint32_t)std::max(0, atoi(t.c_str() + 10));
    else if (t.compare(0, 4, "lib=") == 0) lib = t.substr(4);
    else if (!t.empty()) rest += (rest.empty() ? "" : ",") + t;
  }
  std::unique_ptr<Backend> b;
  if (name == "fake") b = make_fake_backend(rest);
  else if (name == "amdsmi" || name.empty()) b = make_amdsmi_backend(!lib.empty() ? lib : rest);
  if (b && replicate > 1) b = make_replicate_backend(std::move(b), replicate);
  return b;
}

}  // namespace gpumon
