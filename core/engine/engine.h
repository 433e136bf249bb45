// Host engine.  `Engine` is the operation set every front end (C ABI, CLIs,
// exporter, REST, RPC server) is written against; `LocalEngine` implements it
// in-process on top of a Backend, `RemoteEngine` (core/rpc/client.cc) forwards
// it to a gpumon-hostengine daemon.  This is the component the reference gets
// from the closed libdcgm / nv-hostengine pair (bindings/go/dcgm/admin.go:41-209).
#pragma once

#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

#include "../backend/backend.h"
#include "gpumon.h"

namespace gpumon {

using ViolationFn = std::function<void(const gm_violation_t&)>;
using EventFn = std::function<void(const gm_event_t&)>;

class Engine {
 public:
  virtual ~Engine() {}
  virtual bool is_remote() const { return false; }

  virtual gm_status_t device_count(uint32_t* n) = 0;
  virtual gm_status_t supported_devices(std::vector<uint32_t>* ids) = 0;
  virtual gm_status_t device_attributes(uint32_t gpu, gm_device_attributes_t* out) = 0;
  virtual gm_status_t device_topology(uint32_t gpu, gm_topology_t* out) = 0;

  virtual gm_status_t group_create(const std::string& name, gm_group_t* out) = 0;
  virtual gm_status_t group_add(gm_group_t g, uint32_t gpu) = 0;
  virtual gm_status_t group_remove(gm_group_t g, uint32_t gpu) = 0;
  virtual gm_status_t group_devices(gm_group_t g, std::vector<uint32_t>* ids) = 0;
  virtual gm_status_t group_destroy(gm_group_t g) = 0;
  virtual gm_status_t field_group_create(const std::vector<uint16_t>& ids, const std::string& name,
                                         gm_field_group_t* out) = 0;
  virtual gm_status_t field_group_destroy(gm_field_group_t fg) = 0;

  virtual gm_status_t watch_fields(gm_group_t g, gm_field_group_t fg, int64_t freq_us, double keep_age_s,
                                   int32_t keep_samples) = 0;
  virtual gm_status_t unwatch_fields(gm_group_t g, gm_field_group_t fg) = 0;
  virtual gm_status_t update_all(bool wait) = 0;
  virtual gm_status_t latest_values(uint32_t gpu, const uint16_t* ids, int n, gm_value_t* out) = 0;
  virtual gm_status_t values_since(uint32_t gpu, uint16_t id, int64_t since_us, std::vector<gm_value_t>* out) = 0;
  virtual gm_status_t inject_value(uint32_t gpu, const gm_value_t& v, int32_t hold_ms) = 0;
  virtual gm_status_t fake_ctl(const std::string& cmd) = 0;

  virtual gm_status_t device_status(uint32_t gpu, gm_device_status_t* out) = 0;
  virtual gm_status_t device_processes(uint32_t gpu, std::vector<gm_process_t>* out) = 0;
  // Where the process list of this GPU comes from on this host ("kfd", "fdinfo", "open-files", "fake", "none").
  virtual gm_status_t process_source(uint32_t gpu, std::string* out) = 0;
  virtual gm_status_t metrics_table_source(uint32_t gpu, std::string* out) = 0;
  virtual gm_status_t device_mode(uint32_t gpu, gm_device_mode_t* out) = 0;

  virtual gm_status_t health_set(gm_group_t g, uint32_t systems) = 0;
  virtual gm_status_t health_get(gm_group_t g, uint32_t* systems) = 0;
  virtual gm_status_t health_check(gm_group_t g, std::vector<gm_health_response_t>* out) = 0;

  virtual gm_status_t policy_set(gm_group_t g, const gm_policy_t& p) = 0;
  virtual gm_status_t policy_get(gm_group_t g, gm_policy_t* p) = 0;
  virtual gm_status_t policy_register(gm_group_t g, uint32_t conditions, ViolationFn fn, gm_policy_reg_t* out) = 0;
  virtual gm_status_t policy_unregister(gm_policy_reg_t reg) = 0;

  virtual gm_status_t watch_pid_fields(gm_group_t g, int64_t freq_us, double keep_age_s, int32_t keep_samples) = 0;
  virtual gm_status_t pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out) = 0;

  virtual gm_status_t introspect_toggle(bool enabled) = 0;
  virtual gm_status_t introspect(gm_introspect_t* out) = 0;

  virtual gm_status_t event_subscribe(EventFn fn, uint32_t* sub) = 0;
  virtual gm_status_t event_unsubscribe(uint32_t sub) = 0;

  virtual gm_status_t xgmi_links(uint32_t gpu, std::vector<gm_xgmi_link_t>* out) = 0;
  virtual gm_status_t health_thresholds_set(const gm_health_thresholds_t& t) = 0;
  virtual gm_status_t health_thresholds_get(gm_health_thresholds_t* t) = 0;
  virtual gm_status_t diag_submit(const gm_diag_result_t& r) = 0;
  virtual gm_status_t diag_results(uint32_t gpu, std::vector<gm_diag_result_t>* out) = 0;
};

// ----------------------------------------------------------------------------

struct HistoryPoint {
  int64_t ts_us;
  SlotValue v;
};

// Serialises callbacks off the sampler and event threads: a bounded queue
// drained by one notifier thread.  A slow subscriber can only delay other
// notifications, never sampling; overflow drops the oldest task and counts it.
class Notifier {
 public:
  Notifier();
  ~Notifier();
  void post(std::function<void()> task);
  void drain();  // blocks until everything posted so far has run
  uint64_t dropped() const { return dropped_.load(); }

 private:
  void run();
  std::mutex mu_;
  std::condition_variable cv_, idle_cv_;
  std::deque<std::function<void()>> q_;
  bool stop_ = false, busy_ = false;
  std::atomic<uint64_t> dropped_{0};
  std::thread th_;
};

class LocalEngine : public Engine {
 public:
  explicit LocalEngine(std::unique_ptr<Backend> backend);
  ~LocalEngine() override;
  gm_status_t start();
  void stop();
  Backend* backend() { return backend_.get(); }

  gm_status_t device_count(uint32_t* n) override;
  gm_status_t supported_devices(std::vector<uint32_t>* ids) override;
  gm_status_t device_attributes(uint32_t gpu, gm_device_attributes_t* out) override;
  gm_status_t device_topology(uint32_t gpu, gm_topology_t* out) override;
  gm_status_t group_create(const std::string& name, gm_group_t* out) override;
  gm_status_t group_add(gm_group_t g, uint32_t gpu) override;
  gm_status_t group_remove(gm_group_t g, uint32_t gpu) override;
  gm_status_t group_devices(gm_group_t g, std::vector<uint32_t>* ids) override;
  gm_status_t group_destroy(gm_group_t g) override;
  gm_status_t field_group_create(const std::vector<uint16_t>& ids, const std::string& name,
                                 gm_field_group_t* out) override;
  gm_status_t field_group_destroy(gm_field_group_t fg) override;
  gm_status_t watch_fields(gm_group_t g, gm_field_group_t fg, int64_t freq_us, double keep_age_s,
                           int32_t keep_samples) override;
  gm_status_t unwatch_fields(gm_group_t g, gm_field_group_t fg) override;
  gm_status_t update_all(bool wait) override;
  gm_status_t latest_values(uint32_t gpu, const uint16_t* ids, int n, gm_value_t* out) override;
  gm_status_t values_since(uint32_t gpu, uint16_t id, int64_t since_us, std::vector<gm_value_t>* out) override;
  gm_status_t inject_value(uint32_t gpu, const gm_value_t& v, int32_t hold_ms) override;
  gm_status_t fake_ctl(const std::string& cmd) override;
  gm_status_t device_status(uint32_t gpu, gm_device_status_t* out) override;
  gm_status_t device_processes(uint32_t gpu, std::vector<gm_process_t>* out) override;
  gm_status_t process_source(uint32_t gpu, std::string* out) override;
  gm_status_t metrics_table_source(uint32_t gpu, std::string* out) override;
  gm_status_t device_mode(uint32_t gpu, gm_device_mode_t* out) override;
  gm_status_t health_set(gm_group_t g, uint32_t systems) override;
  gm_status_t health_get(gm_group_t g, uint32_t* systems) override;
  gm_status_t health_check(gm_group_t g, std::vector<gm_health_response_t>* out) override;
  gm_status_t policy_set(gm_group_t g, const gm_policy_t& p) override;
  gm_status_t policy_get(gm_group_t g, gm_policy_t* p) override;
  gm_status_t policy_register(gm_group_t g, uint32_t conditions, ViolationFn fn, gm_policy_reg_t* out) override;
  gm_status_t policy_unregister(gm_policy_reg_t reg) override;
  gm_status_t watch_pid_fields(gm_group_t g, int64_t freq_us, double keep_age_s, int32_t keep_samples) override;
  gm_status_t pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out) override;
  gm_status_t introspect_toggle(bool enabled) override;
  gm_status_t introspect(gm_introspect_t* out) override;
  gm_status_t event_subscribe(EventFn fn, uint32_t* sub) override;
  gm_status_t event_unsubscribe(uint32_t sub) override;
  gm_status_t xgmi_links(uint32_t gpu, std::vector<gm_xgmi_link_t>* out) override;
  gm_status_t health_thresholds_set(const gm_health_thresholds_t& t) override;
  gm_status_t health_thresholds_get(gm_health_thresholds_t* t) override;
  gm_status_t diag_submit(const gm_diag_result_t& r) override;
  gm_status_t diag_results(uint32_t gpu, std::vector<gm_diag_result_t>* out) override;

 private:
  // ---- snapshot (seqlock) ---------------------------------------------------
  struct Cell {
    std::atomic<uint64_t> bits{0};
    std::atomic<int64_t> ts{0};
    std::atomic<uint8_t> status{GM_VS_BLANK};
  };
  struct Hold {
    SlotValue v;
    int64_t until_mono_us;  // 0 = until the next real sample
  };
  struct SlotWatch {
    int64_t freq_us = 0;  // 0 = not watched
    int64_t keep_age_us = 0;
    int32_t keep_samples = 0;
    int64_t next_due_mono = 0;
  };
  struct GpuState {
    GpuStatic st;
    std::atomic<uint64_t> seq{0};
    std::mutex write_mu;  // serialises writers; readers never take it
    std::unique_ptr<Cell[]> cells;
    std::vector<SlotWatch> watch;                  // sampler thread + state_mu_
    std::vector<std::deque<HistoryPoint>> history; // hist_mu
    std::mutex hist_mu;
    std::map<int, Hold> holds;                     // hist_mu
    std::vector<int64_t> critical_event_ts;        // hist_mu
    uint32_t last_event_code = 0;
    // RAS events synthesised from counter steps: the totals seen by the previous ECC sample (-1 = none yet)
    int64_t prev_ue = -1, prev_deferred = -1;      // sampler thread only
    std::map<std::string, gm_diag_result_t> diag;  // hist_mu; latest result per test
  };

  struct Watch {
    bool internal = false;
    gm_group_t group = 0;
    gm_field_group_t fg = 0;
    std::vector<uint32_t> gpus;  // internal watches only
    std::vector<uint16_t> fields;
    int64_t freq_us;
    int64_t keep_age_us;
    int32_t keep_samples;
  };
  struct Group {
    std::string name;
    std::vector<uint32_t> gpus;
    uint32_t health_systems = 0;
    bool has_policy = false;
    gm_policy_t policy;
  };
  struct HealthBaseline {
    bool valid = false;
    std::map<uint16_t, int64_t> counters;
    int64_t last_check_us = 0;
    size_t events_seen = 0;
    uint64_t backend_errors = 0;
  };
  struct PolicyArm {  // per (registration, gpu)
    bool thermal_fired = false, power_fired = false, pages_fired = false;
    bool baseline_valid = false;
    int64_t ue = 0, replay = 0, xgmi_ce = 0, xgmi_ue = 0, links_down = 0;
  };
  struct PolicyReg {
    gm_group_t group;
    uint32_t conditions;
    ViolationFn fn;
    std::map<uint32_t, PolicyArm> arm;
  };
  struct ProcAcct {
    uint32_t pid = 0;
    std::string name;
    int64_t start_us = 0, end_us = 0, last_seen_mono = 0;
    double energy_mj = 0;
    double sum_gfx = 0, sum_mem = 0, sum_gfxclk = 0, sum_memclk = 0, sum_share = 0;
    uint64_t nsamples = 0;
    uint64_t max_vram = 0;
    bool have_vram = false;  // false while the process list carries no memory figures
    uint64_t last_engine_ns = 0;
    bool have_engine = false;
    int64_t pcie_rx0 = -1, pcie_tx0 = -1, pcie_rx1 = -1, pcie_tx1 = -1, replay0 = -1, replay1 = -1;
    int64_t ce0 = -1, ce1 = -1, ue0 = -1, ue1 = -1;
    int64_t viol0[5] = {-1, -1, -1, -1, -1}, viol1[5] = {-1, -1, -1, -1, -1};
    std::vector<int64_t> event_ts;
  };
  struct PidWatch {
    int64_t freq_us = 0;
    int64_t keep_age_us = 0;
    int64_t next_due_mono = 0;
    int64_t last_tick_mono = 0;
    std::map<uint32_t, ProcAcct> procs;  // by pid
  };

  void sampler_loop();
  void event_loop();
  // due[slot] != 0: append to the slot's history; retention comes from `watch`, the
  // sampler's own copy of the watch table taken under state_mu_.
  void sample_gpu(uint32_t gpu, uint32_t sources, const std::vector<uint8_t>& due, const std::vector<SlotWatch>& watch,
                  int64_t now_mono, int64_t now_wall);
  void publish(GpuState& g, int slot, const SlotValue& v, int64_t ts);
  bool read_cell(GpuState& g, int slot, SlotValue* v, int64_t* ts);
  void rebuild_watch_table();  // state_mu_ held
  gm_status_t resolve_group(gm_group_t g, std::vector<uint32_t>* gpus);  // state_mu_ held
  void ensure_internal_watch(const std::string& key, const std::vector<uint32_t>& gpus,
                             const std::vector<uint16_t>& fields, int64_t freq_us);  // state_mu_ held
  void evaluate_policies(int64_t now_wall);
  void accounting_tick(int64_t now_mono, int64_t now_wall);
  void handle_event(const DeviceEvent& e);
  void fire(const std::shared_ptr<PolicyReg>& reg, const gm_violation_t& v);
  bool get_i(uint32_t gpu, uint16_t id, int64_t* out);
  bool get_d(uint32_t gpu, uint16_t id, double* out);

  std::unique_ptr<Backend> backend_;
  std::vector<std::unique_ptr<GpuState>> gpus_;
  std::atomic<bool> running_{false};

  std::mutex state_mu_;  // groups, field groups, watches, policy/health tables
  std::map<gm_group_t, Group> groups_;
  std::map<gm_field_group_t, std::vector<uint16_t>> field_groups_;
  std::map<gm_field_group_t, std::string> field_group_names_;
  std::vector<Watch> watches_;
  std::map<std::string, Watch> internal_watches_;  // engine-owned: status, health, policy, accounting
  uint32_t next_group_ = 1, next_fg_ = 1, next_reg_ = 1, next_sub_ = 1;
  std::map<std::pair<gm_group_t, uint32_t>, HealthBaseline> health_base_;
  std::map<gm_policy_reg_t, std::shared_ptr<PolicyReg>> policy_regs_;
  std::map<uint32_t, PidWatch> pid_watch_;  // by gpu
  std::map<uint32_t, EventFn> event_subs_;
  gm_health_thresholds_t thresholds_;

  std::mutex wake_mu_;
  std::condition_variable wake_cv_, cycle_cv_;
  bool force_ = false;
  uint64_t cycles_ = 0;
  uint64_t force_requested_ = 0;
  std::atomic<uint64_t> watch_gen_{0};

  std::thread sampler_, events_;
  Notifier notifier_;

  // introspection
  std::atomic<bool> introspect_on_{false};
  int64_t intro_t0_mono_ = 0;
  double intro_cpu0_ = 0;
  int64_t start_mono_ = 0;
  std::atomic<uint64_t> backend_errors_{0};
  std::atomic<int64_t> last_cycle_us_{0};
  std::atomic<int64_t> total_cycle_us_{0};
  std::atomic<uint64_t> timed_cycles_{0};
};

// Identifier helpers shared by engine, CLI and REST.
const char* link_type_name(int32_t link, uint32_t hops);
const char* link_type_description(int32_t link);
const char* health_system_name(uint32_t system);
const char* health_result_name(int32_t health);
const char* policy_condition_name(uint32_t condition);
const char* event_code_name(uint32_t code);
const char* status_string(gm_status_t st);
gm_health_thresholds_t health_default_thresholds();
void fill_attributes(const GpuStatic& s, gm_device_attributes_t* out);
// Value of a static (read-once) field out of the attribute cache.
void static_value(const GpuStatic& s, uint16_t id, gm_value_t* out);
// The fields behind gm_device_status_t, in GM_DS_* bit order, and the struct fill from their values.
extern const uint16_t kStatusFields[];
extern const int kNumStatusFields;
void fill_status(uint32_t gpu, const gm_value_t* vals, gm_device_status_t* out);

}  // namespace gpumon
