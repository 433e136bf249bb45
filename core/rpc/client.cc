// RemoteEngine: the Engine interface over a socket to gpumon-hostengine.  One
// reader thread demultiplexes replies (by correlation id) and asynchronous
// notifications (policy violations, driver events); callbacks run on a
// dedicated delivery thread so a slow subscriber never stalls replies.
#include <condition_variable>
#include <sys/socket.h>
#include <unistd.h>

#include "../common/util.h"
#include "server.h"
#include "wire.h"

namespace gpumon {
namespace {

class RemoteEngine : public Engine {
 public:
  RemoteEngine(int fd) : fd_(fd) {
    reader_ = std::thread([this] { read_loop(); });
  }
  ~RemoteEngine() override {
    closing_ = true;
    shutdown(fd_, SHUT_RDWR);
    if (reader_.joinable()) reader_.join();
    close(fd_);
  }
  bool is_remote() const override { return true; }

  gm_status_t hello() {
    WireWriter w;
    w.u32(kProtocolVersion);
    std::string rep;
    gm_status_t st = call(OP_HELLO, w, &rep);
    return st;
  }

  gm_status_t device_count(uint32_t* n) override {
    WireWriter w;
    std::string rep;
    gm_status_t st = call(OP_DEVICE_COUNT, w, &rep);
    WireReader r(rep.data(), rep.size());
    *n = r.u32();
    return r.ok() ? st : GM_ST_CONNECTION_NOT_VALID;
  }
  gm_status_t supported_devices(std::vector<uint32_t>* ids) override {
    WireWriter w;
    std::string rep;
    gm_status_t st = call(OP_SUPPORTED_DEVICES, w, &rep);
    return read_u32_list(rep, st, ids);
  }
  gm_status_t device_attributes(uint32_t gpu, gm_device_attributes_t* out) override {
    if (out->version != GM_DEVICE_ATTRIBUTES_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    w.u32(gpu);
    return call_pod(OP_DEVICE_ATTRIBUTES, w, out);
  }
  gm_status_t device_topology(uint32_t gpu, gm_topology_t* out) override {
    if (out->version != GM_TOPOLOGY_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    w.u32(gpu);
    return call_pod(OP_DEVICE_TOPOLOGY, w, out);
  }
  gm_status_t group_create(const std::string& name, gm_group_t* out) override {
    WireWriter w;
    w.str(name);
    return call_u32(OP_GROUP_CREATE, w, out);
  }
  gm_status_t group_add(gm_group_t g, uint32_t gpu) override {
    WireWriter w;
    w.u32(g); w.u32(gpu);
    return call(OP_GROUP_ADD, w, nullptr);
  }
  gm_status_t group_remove(gm_group_t g, uint32_t gpu) override {
    WireWriter w;
    w.u32(g); w.u32(gpu);
    return call(OP_GROUP_REMOVE, w, nullptr);
  }
  gm_status_t group_devices(gm_group_t g, std::vector<uint32_t>* ids) override {
    WireWriter w;
    w.u32(g);
    std::string rep;
    gm_status_t st = call(OP_GROUP_DEVICES, w, &rep);
    return read_u32_list(rep, st, ids);
  }
  gm_status_t group_destroy(gm_group_t g) override {
    WireWriter w;
    w.u32(g);
    return call(OP_GROUP_DESTROY, w, nullptr);
  }
  gm_status_t field_group_create(const std::vector<uint16_t>& ids, const std::string& name,
                                 gm_field_group_t* out) override {
    WireWriter w;
    w.u32((uint32_t)ids.size());
    for (auto i : ids) w.u32(i);
    w.str(name);
    return call_u32(OP_FG_CREATE, w, out);
  }
  gm_status_t field_group_destroy(gm_field_group_t fg) override {
    WireWriter w;
    w.u32(fg);
    return call(OP_FG_DESTROY, w, nullptr);
  }
  gm_status_t watch_fields(gm_group_t g, gm_field_group_t fg, int64_t freq_us, double keep_age_s,
                           int32_t keep_samples) override {
    WireWriter w;
    w.u32(g); w.u32(fg); w.i64(freq_us); w.f64(keep_age_s); w.i32(keep_samples);
    return call(OP_WATCH_FIELDS, w, nullptr);
  }
  gm_status_t unwatch_fields(gm_group_t g, gm_field_group_t fg) override {
    WireWriter w;
    w.u32(g); w.u32(fg);
    return call(OP_UNWATCH_FIELDS, w, nullptr);
  }
  gm_status_t update_all(bool wait) override {
    WireWriter w;
    w.u32(wait ? 1 : 0);
    return call(OP_UPDATE_ALL, w, nullptr);
  }
  gm_status_t latest_values(uint32_t gpu, const uint16_t* ids, int n, gm_value_t* out) override {
    WireWriter w;
    w.u32(gpu); w.u32((uint32_t)n);
    for (int i = 0; i < n; i++) w.u32(ids[i]);
    std::string rep;
    gm_status_t st = call(OP_LATEST_VALUES, w, &rep);
    if (st != GM_ST_OK) return st;
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32();
    if (cnt != (uint32_t)n) return GM_ST_CONNECTION_NOT_VALID;
    r.raw(out, (size_t)n * sizeof(gm_value_t));
    return r.ok() ? GM_ST_OK : GM_ST_VER_MISMATCH;
  }
  gm_status_t values_since(uint32_t gpu, uint16_t id, int64_t since_us, std::vector<gm_value_t>* out) override {
    WireWriter w;
    w.u32(gpu); w.u32(id); w.i64(since_us);
    std::string rep;
    gm_status_t st = call(OP_VALUES_SINCE, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32();
    out->clear();
    if (!r.ok() || cnt > kMaxFrame / sizeof(gm_value_t)) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    out->resize(cnt);
    if (cnt) r.raw(out->data(), cnt * sizeof(gm_value_t));
    return r.ok() ? st : GM_ST_VER_MISMATCH;
  }
  gm_status_t inject_value(uint32_t gpu, const gm_value_t& v, int32_t hold_ms) override {
    WireWriter w;
    w.u32(gpu); w.pod(v); w.i32(hold_ms);
    return call(OP_INJECT_VALUE, w, nullptr);
  }
  gm_status_t fake_ctl(const std::string& cmd) override {
    WireWriter w;
    w.str(cmd);
    return call(OP_FAKE_CTL, w, nullptr);
  }
  gm_status_t device_status(uint32_t gpu, gm_device_status_t* out) override {
    if (out->version != GM_DEVICE_STATUS_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    w.u32(gpu);
    return call_pod(OP_DEVICE_STATUS, w, out);
  }
  gm_status_t device_processes(uint32_t gpu, std::vector<gm_process_t>* out) override {
    WireWriter w;
    w.u32(gpu);
    std::string rep;
    gm_status_t st = call(OP_DEVICE_PROCESSES, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32();
    out->clear();
    if (!r.ok() || cnt > 65536) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    out->resize(cnt);
    if (cnt) r.raw(out->data(), cnt * sizeof(gm_process_t));
    return r.ok() ? st : GM_ST_VER_MISMATCH;
  }
  gm_status_t process_source(uint32_t gpu, std::string* out) override {
    WireWriter w;
    w.u32(gpu);
    std::string rep;
    gm_status_t st = call(OP_PROCESS_SOURCE, w, &rep);
    WireReader r(rep.data(), rep.size());
    *out = r.str();
    return r.ok() ? st : (st != GM_ST_OK ? st : GM_ST_VER_MISMATCH);
  }
  gm_status_t metrics_table_source(uint32_t gpu, std::string* out) override {
    WireWriter w;
    w.u32(gpu);
    std::string rep;
    gm_status_t st = call(OP_METRICS_TABLE_SOURCE, w, &rep);
    WireReader r(rep.data(), rep.size());
    *out = r.str();
    return r.ok() ? st : (st != GM_ST_OK ? st : GM_ST_VER_MISMATCH);
  }
  gm_status_t device_mode(uint32_t gpu, gm_device_mode_t* out) override {
    if (out->version != GM_DEVICE_MODE_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    w.u32(gpu);
    return call_pod(OP_DEVICE_MODE, w, out);
  }
  gm_status_t health_set(gm_group_t g, uint32_t systems) override {
    WireWriter w;
    w.u32(g); w.u32(systems);
    return call(OP_HEALTH_SET, w, nullptr);
  }
  gm_status_t health_get(gm_group_t g, uint32_t* systems) override {
    WireWriter w;
    w.u32(g);
    return call_u32(OP_HEALTH_GET, w, systems);
  }
  gm_status_t health_check(gm_group_t g, std::vector<gm_health_response_t>* out) override {
    WireWriter w;
    w.u32(g);
    std::string rep;
    gm_status_t st = call(OP_HEALTH_CHECK, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32(), sz = r.u32();
    out->clear();
    if (!r.ok()) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    if (sz != sizeof(gm_health_response_t) || cnt > GM_MAX_GPUS) return GM_ST_VER_MISMATCH;
    out->resize(cnt);
    if (cnt) r.raw(out->data(), cnt * sizeof(gm_health_response_t));
    return r.ok() ? st : GM_ST_VER_MISMATCH;
  }
  template <typename T>
  gm_status_t call_list(uint32_t op, const WireWriter& w, std::vector<T>* out, uint32_t max_count) {
    std::string rep;
    gm_status_t st = call(op, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32(), sz = r.u32();
    out->clear();
    if (!r.ok()) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    if (sz != sizeof(T) || cnt > max_count) return GM_ST_VER_MISMATCH;
    out->resize(cnt);
    if (cnt) r.raw(out->data(), cnt * sizeof(T));
    return r.ok() ? st : GM_ST_VER_MISMATCH;
  }
  gm_status_t xgmi_links(uint32_t gpu, std::vector<gm_xgmi_link_t>* out) override {
    WireWriter w;
    w.u32(gpu);
    return call_list(OP_XGMI_LINKS, w, out, 64);
  }
  gm_status_t health_thresholds_set(const gm_health_thresholds_t& t) override {
    WireWriter w;
    w.pod(t);
    return call(OP_HEALTH_THRESHOLDS_SET, w, nullptr);
  }
  gm_status_t health_thresholds_get(gm_health_thresholds_t* t) override {
    if (t->version != GM_HEALTH_THRESHOLDS_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    return call_pod(OP_HEALTH_THRESHOLDS_GET, w, t);
  }
  gm_status_t diag_submit(const gm_diag_result_t& d) override {
    WireWriter w;
    w.pod(d);
    return call(OP_DIAG_SUBMIT, w, nullptr);
  }
  gm_status_t diag_results(uint32_t gpu, std::vector<gm_diag_result_t>* out) override {
    WireWriter w;
    w.u32(gpu);
    return call_list(OP_DIAG_RESULTS, w, out, 256);
  }
  gm_status_t policy_set(gm_group_t g, const gm_policy_t& p) override {
    WireWriter w;
    w.u32(g); w.pod(p);
    return call(OP_POLICY_SET, w, nullptr);
  }
  gm_status_t policy_get(gm_group_t g, gm_policy_t* p) override {
    WireWriter w;
    w.u32(g);
    return call_pod(OP_POLICY_GET, w, p);
  }
  gm_status_t policy_register(gm_group_t g, uint32_t conditions, ViolationFn fn, gm_policy_reg_t* out) override {
    uint32_t local;
    {
      std::lock_guard<std::mutex> l(cb_mu_);
      local = next_local_++;
      violation_fns_[local] = std::move(fn);
    }
    WireWriter w;
    w.u32(g); w.u32(conditions); w.u32(local);
    uint32_t remote = 0;
    gm_status_t st = call_u32(OP_POLICY_REGISTER, w, &remote);
    std::lock_guard<std::mutex> l(cb_mu_);
    if (st != GM_ST_OK) {
      violation_fns_.erase(local);
      return st;
    }
    reg_remote_[local] = remote;
    *out = local;
    return GM_ST_OK;
  }
  gm_status_t policy_unregister(gm_policy_reg_t reg) override {
    uint32_t remote;
    {
      std::lock_guard<std::mutex> l(cb_mu_);
      auto it = reg_remote_.find(reg);
      if (it == reg_remote_.end()) return GM_ST_NOT_FOUND;
      remote = it->second;
      reg_remote_.erase(it);
      violation_fns_.erase(reg);
    }
    WireWriter w;
    w.u32(remote);
    return call(OP_POLICY_UNREGISTER, w, nullptr);
  }
  gm_status_t watch_pid_fields(gm_group_t g, int64_t freq_us, double keep_age_s, int32_t keep_samples) override {
    WireWriter w;
    w.u32(g); w.i64(freq_us); w.f64(keep_age_s); w.i32(keep_samples);
    return call(OP_WATCH_PID_FIELDS, w, nullptr);
  }
  gm_status_t pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out) override {
    if (out->version != GM_PID_INFO_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    w.u32(g); w.u32(pid);
    std::string rep;
    gm_status_t st = call(OP_PID_INFO, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t cnt = r.u32(), sz = r.u32();
    uint32_t ver = out->version;
    memset(out, 0, sizeof *out);
    out->version = ver;
    out->pid = pid;
    if (!r.ok()) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    if (sz != sizeof(gm_pid_gpu_stats_t) || cnt > GM_MAX_GPUS) return GM_ST_VER_MISMATCH;
    out->num_gpus = cnt;
    if (cnt) r.raw(out->gpus, cnt * sizeof(gm_pid_gpu_stats_t));
    return r.ok() ? st : GM_ST_VER_MISMATCH;
  }
  gm_status_t introspect_toggle(bool enabled) override {
    WireWriter w;
    w.u32(enabled ? 1 : 0);
    return call(OP_INTROSPECT_TOGGLE, w, nullptr);
  }
  gm_status_t introspect(gm_introspect_t* out) override {
    if (out->version != GM_INTROSPECT_VERSION) return GM_ST_VER_MISMATCH;
    WireWriter w;
    return call_pod(OP_INTROSPECT, w, out);
  }
  gm_status_t event_subscribe(EventFn fn, uint32_t* sub) override {
    uint32_t local;
    {
      std::lock_guard<std::mutex> l(cb_mu_);
      local = next_local_++;
      event_fns_[local] = std::move(fn);
    }
    WireWriter w;
    w.u32(local);
    uint32_t remote = 0;
    gm_status_t st = call_u32(OP_EVENT_SUBSCRIBE, w, &remote);
    std::lock_guard<std::mutex> l(cb_mu_);
    if (st != GM_ST_OK) {
      event_fns_.erase(local);
      return st;
    }
    sub_remote_[local] = remote;
    *sub = local;
    return GM_ST_OK;
  }
  gm_status_t event_unsubscribe(uint32_t sub) override {
    uint32_t remote;
    {
      std::lock_guard<std::mutex> l(cb_mu_);
      auto it = sub_remote_.find(sub);
      if (it == sub_remote_.end()) return GM_ST_NOT_FOUND;
      remote = it->second;
      sub_remote_.erase(it);
      event_fns_.erase(sub);
    }
    WireWriter w;
    w.u32(remote);
    return call(OP_EVENT_UNSUBSCRIBE, w, nullptr);
  }

 private:
  struct Pending {
    bool done = false;
    std::string payload;
  };

  // Sends a request and waits for its reply; *rest receives the reply after the status word.
  gm_status_t call(uint32_t op, const WireWriter& w, std::string* rest) {
    uint32_t id;
    {
      std::unique_lock<std::mutex> l(mu_);
      if (dead_) return GM_ST_CONNECTION_NOT_VALID;
      id = next_id_++;
      pending_[id] = Pending();
    }
    {
      std::lock_guard<std::mutex> l(write_mu_);
      if (!rpc_send_frame(fd_, op, id, w.data())) {
        std::lock_guard<std::mutex> l2(mu_);
        pending_.erase(id);
        return GM_ST_CONNECTION_NOT_VALID;
      }
    }
    std::unique_lock<std::mutex> l(mu_);
    bool ok = cv_wait_for(cv_, l, std::chrono::seconds(60), [&] { return pending_[id].done || dead_; });
    Pending p = std::move(pending_[id]);
    pending_.erase(id);
    if (!ok) return GM_ST_TIMEOUT;
    if (!p.done) return GM_ST_CONNECTION_NOT_VALID;
    if (p.payload.size() < 4) return GM_ST_CONNECTION_NOT_VALID;
    int32_t st;
    memcpy(&st, p.payload.data(), 4);
    if (rest) rest->assign(p.payload, 4, std::string::npos);
    return (gm_status_t)st;
  }
  gm_status_t call_u32(uint32_t op, const WireWriter& w, uint32_t* out) {
    std::string rep;
    gm_status_t st = call(op, w, &rep);
    WireReader r(rep.data(), rep.size());
    uint32_t v = r.u32();
    if (r.ok()) *out = v;
    return r.ok() || st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
  }
  template <typename T>
  gm_status_t call_pod(uint32_t op, const WireWriter& w, T* out) {
    std::string rep;
    gm_status_t st = call(op, w, &rep);
    if (rep.empty()) return st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
    WireReader r(rep.data(), rep.size());
    if (!r.pod(out)) return st != GM_ST_OK ? st : GM_ST_VER_MISMATCH;
    return st;
  }
  gm_status_t read_u32_list(const std::string& rep, gm_status_t st, std::vector<uint32_t>* ids) {
    WireReader r(rep.data(), rep.size());
    uint32_t n = r.u32();
    ids->clear();
    for (uint32_t i = 0; i < n && r.ok(); i++) ids->push_back(r.u32());
    return r.ok() || st != GM_ST_OK ? st : GM_ST_CONNECTION_NOT_VALID;
  }

  void read_loop() {
    FrameHeader h;
    std::string payload;
    while (true) {
      int rc = rpc_recv_frame(fd_, &h, &payload);
      if (rc <= 0) break;
      if (h.op == OP_NOTIFY_VIOLATION) {
        gm_violation_t v;
        WireReader r(payload.data(), payload.size());
        if (!r.pod(&v)) continue;
        ViolationFn fn;
        {
          std::lock_guard<std::mutex> l(cb_mu_);
          auto it = violation_fns_.find(h.id);
          if (it != violation_fns_.end()) fn = it->second;
        }
        if (fn) delivery_.post([fn, v] { fn(v); });
      } else if (h.op == OP_NOTIFY_EVENT) {
        gm_event_t e;
        WireReader r(payload.data(), payload.size());
        if (!r.pod(&e)) continue;
        EventFn fn;
        {
          std::lock_guard<std::mutex> l(cb_mu_);
          auto it = event_fns_.find(h.id);
          if (it != event_fns_.end()) fn = it->second;
        }
        if (fn) delivery_.post([fn, e] { fn(e); });
      } else if (h.op & kReplyBit) {
        std::lock_guard<std::mutex> l(mu_);
        auto it = pending_.find(h.id);
        if (it != pending_.end()) {
          it->second.done = true;
          it->second.payload = payload;
          cv_.notify_all();
        }
      }
    }
    std::lock_guard<std::mutex> l(mu_);
    dead_ = true;
    if (!closing_) GM_LOGW("rpc: connection to the host engine lost");
    cv_.notify_all();
  }

  int fd_;
  std::atomic<bool> closing_{false};
  std::thread reader_;
  std::mutex mu_, write_mu_, cb_mu_;
  std::condition_variable cv_;
  std::map<uint32_t, Pending> pending_;
  uint32_t next_id_ = 1, next_local_ = 1;
  bool dead_ = false;
  std::map<uint32_t, ViolationFn> violation_fns_;
  std::map<uint32_t, EventFn> event_fns_;
  std::map<uint32_t, uint32_t> reg_remote_, sub_remote_;
  Notifier delivery_;
};

}  // namespace

std::unique_ptr<Engine> make_remote_engine(const std::string& address, bool is_unix, int timeout_ms,
                                           gm_status_t* status, std::string* error) {
  int fd = rpc_connect(address, is_unix, timeout_ms, error);
  if (fd < 0) {
    *status = GM_ST_CONNECTION_NOT_VALID;
    return nullptr;
  }
  std::unique_ptr<RemoteEngine> e(new RemoteEngine(fd));
  gm_status_t st = e->hello();
  if (st != GM_ST_OK) {
    *status = st;
    *error = "host engine handshake failed";
    return nullptr;
  }
  *status = GM_ST_OK;
  return std::unique_ptr<Engine>(e.release());
}

}  // namespace gpumon
