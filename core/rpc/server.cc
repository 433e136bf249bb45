#include "server.h"

#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include "../common/util.h"
#include "wire.h"

namespace gpumon {

RpcServer::RpcServer(Engine* engine) : engine_(engine) {}
RpcServer::~RpcServer() { stop(); }

gm_status_t RpcServer::start(const std::string& address, bool is_unix) {
  std::string err;
  listen_fd_ = rpc_listen(address, is_unix, &err);
  if (listen_fd_ < 0) {
    GM_LOGE("rpc: %s", err.c_str());
    return GM_ST_CONNECTION_NOT_VALID;
  }
  address_ = address;
  is_unix_ = is_unix;
  running_ = true;
  accept_th_ = std::thread([this] { accept_loop(); });
  GM_LOGI("rpc: serving on %s%s", is_unix ? "unix:" : "", address.c_str());
  return GM_ST_OK;
}

void RpcServer::stop() {
  if (!running_.exchange(false)) return;
  shutdown(listen_fd_, SHUT_RDWR);
  if (accept_th_.joinable()) accept_th_.join();
  close(listen_fd_);
  listen_fd_ = -1;
  if (is_unix_) unlink(address_.c_str());
  std::vector<std::shared_ptr<Client>> cs;
  {
    std::lock_guard<std::mutex> l(mu_);
    cs.swap(clients_);
  }
  for (auto& c : cs) {
    {
      // serve() closes the descriptor under write_mu when its peer leaves first;
      // never shut down a number that may already belong to someone else.
      std::lock_guard<std::mutex> l(c->write_mu);
      if (c->fd >= 0) shutdown(c->fd, SHUT_RDWR);
    }
    if (c->th.joinable()) c->th.join();
  }
}

void RpcServer::accept_loop() {
  while (running_) {
    pollfd p = {listen_fd_, POLLIN, 0};
    int rc = poll(&p, 1, 500);
    if (rc <= 0) continue;
    int fd = accept(listen_fd_, nullptr, nullptr);
    if (fd < 0) continue;
    auto c = std::make_shared<Client>();
    c->fd = fd;
    std::lock_guard<std::mutex> l(mu_);
    // Reap finished connections.
    for (auto it = clients_.begin(); it != clients_.end();) {
      if (!(*it)->alive) {
        if ((*it)->th.joinable()) (*it)->th.join();
        it = clients_.erase(it);
      } else {
        ++it;
      }
    }
    c->th = std::thread([this, c] { serve(c); });
    clients_.push_back(c);
  }
}

void RpcServer::serve(std::shared_ptr<Client> c) {
  FrameHeader h;
  std::string payload;
  while (running_) {
    int rc = rpc_recv_frame(c->fd, &h, &payload);
    if (rc <= 0) break;
    std::string reply = dispatch(c, h.op, payload);
    std::lock_guard<std::mutex> l(c->write_mu);
    if (!rpc_send_frame(c->fd, h.op | kReplyBit, h.id, reply)) break;
  }
  c->alive = false;
  // Release what the connection owned.
  for (auto r : c->regs) engine_->policy_unregister(r);
  for (auto s : c->subs) engine_->event_unsubscribe(s);
  for (auto fg : c->field_groups) engine_->field_group_destroy(fg);
  for (auto g : c->groups) engine_->group_destroy(g);
  {
    std::lock_guard<std::mutex> l(c->write_mu);
    close(c->fd);
    c->fd = -1;
  }
}

std::string RpcServer::dispatch(const std::shared_ptr<Client>& c, uint32_t op, const std::string& req) {
  WireReader r(req.data(), req.size());
  WireWriter w;
  auto status_only = [&](gm_status_t st) {
    WireWriter o;
    o.i32(st);
    return o.data();
  };
  switch (op) {
    case OP_HELLO: {
      uint32_t ver = r.u32();
      w.i32(ver == kProtocolVersion ? GM_ST_OK : GM_ST_VER_MISMATCH);
      w.u32(kProtocolVersion);
      return w.data();
    }
    case OP_DEVICE_COUNT: {
      uint32_t n = 0;
      w.i32(engine_->device_count(&n));
      w.u32(n);
      return w.data();
    }
    case OP_SUPPORTED_DEVICES: {
      std::vector<uint32_t> ids;
      w.i32(engine_->supported_devices(&ids));
      w.u32((uint32_t)ids.size());
      for (auto i : ids) w.u32(i);
      return w.data();
    }
    case OP_DEVICE_ATTRIBUTES: {
      uint32_t gpu = r.u32();
      gm_device_attributes_t a;
      memset(&a, 0, sizeof a);  // an error return leaves it untouched: never send stack bytes
      a.version = GM_DEVICE_ATTRIBUTES_VERSION;
      w.i32(engine_->device_attributes(gpu, &a));
      w.pod(a);
      return w.data();
    }
    case OP_DEVICE_TOPOLOGY: {
      uint32_t gpu = r.u32();
      gm_topology_t t;
      memset(&t, 0, sizeof t);  // an error return leaves it untouched: never send stack bytes
      t.version = GM_TOPOLOGY_VERSION;
      w.i32(engine_->device_topology(gpu, &t));
      w.pod(t);
      return w.data();
    }
    case OP_GROUP_CREATE: {
      std::string name = r.str();
      gm_group_t g = 0;
      gm_status_t st = engine_->group_create(name, &g);
      if (st == GM_ST_OK) c->groups.insert(g);
      w.i32(st);
      w.u32(g);
      return w.data();
    }
    case OP_GROUP_ADD: {
      uint32_t g = r.u32(), gpu = r.u32();
      return status_only(engine_->group_add(g, gpu));
    }
    case OP_GROUP_REMOVE: {
      uint32_t g = r.u32(), gpu = r.u32();
      return status_only(engine_->group_remove(g, gpu));
    }
    case OP_GROUP_DEVICES: {
      uint32_t g = r.u32();
      std::vector<uint32_t> ids;
      w.i32(engine_->group_devices(g, &ids));
      w.u32((uint32_t)ids.size());
      for (auto i : ids) w.u32(i);
      return w.data();
    }
    case OP_GROUP_DESTROY: {
      uint32_t g = r.u32();
      c->groups.erase(g);
      return status_only(engine_->group_destroy(g));
    }
    case OP_FG_CREATE: {
      uint32_t n = r.u32();
      if (n > GM_MAX_FIELDS_PER_GROUP) return status_only(GM_ST_BADPARAM);
      std::vector<uint16_t> ids(n);
      for (auto& i : ids) i = (uint16_t)r.u32();
      std::string name = r.str();
      if (!r.ok()) return status_only(GM_ST_BADPARAM);
      gm_field_group_t fg = 0;
      gm_status_t st = engine_->field_group_create(ids, name, &fg);
      if (st == GM_ST_OK) c->field_groups.insert(fg);
      w.i32(st);
      w.u32(fg);
      return w.data();
    }
    case OP_FG_DESTROY: {
      uint32_t fg = r.u32();
      c->field_groups.erase(fg);
      return status_only(engine_->field_group_destroy(fg));
    }
    case OP_WATCH_FIELDS: {
      uint32_t g = r.u32(), fg = r.u32();
      int64_t freq = r.i64();
      double age = r.f64();
      int32_t keep = r.i32();
      if (!r.ok()) return status_only(GM_ST_BADPARAM);
      return status_only(engine_->watch_fields(g, fg, freq, age, keep));
    }
    case OP_UNWATCH_FIELDS: {
      uint32_t g = r.u32(), fg = r.u32();
      return status_only(engine_->unwatch_fields(g, fg));
    }
    case OP_UPDATE_ALL: {
      uint32_t wait = r.u32();
      return status_only(engine_->update_all(wait != 0));
    }
    case OP_LATEST_VALUES: {
      uint32_t gpu = r.u32(), n = r.u32();
      if (n > 4096) return status_only(GM_ST_BADPARAM);
      std::vector<uint16_t> ids(n);
      for (auto& i : ids) i = (uint16_t)r.u32();
      if (!r.ok()) return status_only(GM_ST_BADPARAM);
      std::vector<gm_value_t> vals(n);
      gm_status_t st = engine_->latest_values(gpu, ids.data(), (int)n, vals.data());
      w.i32(st);
      w.u32(st == GM_ST_OK ? n : 0);
      if (st == GM_ST_OK && n) w.raw(vals.data(), n * sizeof(gm_value_t));
      return w.data();
    }
    case OP_VALUES_SINCE: {
      uint32_t gpu = r.u32(), id = r.u32();
      int64_t since = r.i64();
      std::vector<gm_value_t> vals;
      gm_status_t st = engine_->values_since(gpu, (uint16_t)id, since, &vals);
      w.i32(st);
      w.u32((uint32_t)vals.size());
      if (!vals.empty()) w.raw(vals.data(), vals.size() * sizeof(gm_value_t));
      return w.data();
    }
    case OP_INJECT_VALUE: {
      uint32_t gpu = r.u32();
      gm_value_t v;
      if (!r.pod(&v)) return status_only(GM_ST_VER_MISMATCH);
      int32_t hold = r.i32();
      return status_only(engine_->inject_value(gpu, v, hold));
    }
    case OP_FAKE_CTL: return status_only(engine_->fake_ctl(r.str()));
    case OP_DEVICE_STATUS: {
      uint32_t gpu = r.u32();
      gm_device_status_t d;
      memset(&d, 0, sizeof d);  // an error return leaves it untouched: never send stack bytes
      d.version = GM_DEVICE_STATUS_VERSION;
      w.i32(engine_->device_status(gpu, &d));
      w.pod(d);
      return w.data();
    }
    case OP_DEVICE_PROCESSES: {
      uint32_t gpu = r.u32();
      std::vector<gm_process_t> ps;
      w.i32(engine_->device_processes(gpu, &ps));
      w.u32((uint32_t)ps.size());
      if (!ps.empty()) w.raw(ps.data(), ps.size() * sizeof(gm_process_t));
      return w.data();
    }
    case OP_PROCESS_SOURCE: {
      uint32_t gpu = r.u32();
      std::string src;
      w.i32(engine_->process_source(gpu, &src));
      w.str(src);
      return w.data();
    }
    case OP_METRICS_TABLE_SOURCE: {
      uint32_t gpu = r.u32();
      std::string src;
      w.i32(engine_->metrics_table_source(gpu, &src));
      w.str(src);
      return w.data();
    }
    case OP_DEVICE_MODE: {
      uint32_t gpu = r.u32();
      gm_device_mode_t m;
      memset(&m, 0, sizeof m);  // an error return leaves it untouched: never send stack bytes
      m.version = GM_DEVICE_MODE_VERSION;
      w.i32(engine_->device_mode(gpu, &m));
      w.pod(m);
      return w.data();
    }
    case OP_HEALTH_SET: {
      uint32_t g = r.u32(), sys = r.u32();
      return status_only(engine_->health_set(g, sys));
    }
    case OP_HEALTH_GET: {
      uint32_t g = r.u32(), sys = 0;
      w.i32(engine_->health_get(g, &sys));
      w.u32(sys);
      return w.data();
    }
    case OP_HEALTH_CHECK: {
      uint32_t g = r.u32();
      std::vector<gm_health_response_t> out;
      w.i32(engine_->health_check(g, &out));
      w.u32((uint32_t)out.size());
      w.u32((uint32_t)sizeof(gm_health_response_t));
      if (!out.empty()) w.raw(out.data(), out.size() * sizeof(gm_health_response_t));
      return w.data();
    }
    case OP_XGMI_LINKS: {
      uint32_t gpu = r.u32();
      std::vector<gm_xgmi_link_t> out;
      w.i32(engine_->xgmi_links(gpu, &out));
      w.u32((uint32_t)out.size());
      w.u32((uint32_t)sizeof(gm_xgmi_link_t));
      if (!out.empty()) w.raw(out.data(), out.size() * sizeof(gm_xgmi_link_t));
      return w.data();
    }
    case OP_HEALTH_THRESHOLDS_SET: {
      gm_health_thresholds_t t;
      if (!r.pod(&t)) return status_only(GM_ST_VER_MISMATCH);
      return status_only(engine_->health_thresholds_set(t));
    }
    case OP_HEALTH_THRESHOLDS_GET: {
      gm_health_thresholds_t t;
      memset(&t, 0, sizeof t);
      t.version = GM_HEALTH_THRESHOLDS_VERSION;
      w.i32(engine_->health_thresholds_get(&t));
      w.pod(t);
      return w.data();
    }
    case OP_DIAG_SUBMIT: {
      gm_diag_result_t d;
      if (!r.pod(&d)) return status_only(GM_ST_VER_MISMATCH);
      return status_only(engine_->diag_submit(d));
    }
    case OP_DIAG_RESULTS: {
      uint32_t gpu = r.u32();
      std::vector<gm_diag_result_t> out;
      w.i32(engine_->diag_results(gpu, &out));
      w.u32((uint32_t)out.size());
      w.u32((uint32_t)sizeof(gm_diag_result_t));
      if (!out.empty()) w.raw(out.data(), out.size() * sizeof(gm_diag_result_t));
      return w.data();
    }
    case OP_POLICY_SET: {
      uint32_t g = r.u32();
      gm_policy_t p;
      if (!r.pod(&p)) return status_only(GM_ST_VER_MISMATCH);
      return status_only(engine_->policy_set(g, p));
    }
    case OP_POLICY_GET: {
      uint32_t g = r.u32();
      gm_policy_t p;
      memset(&p, 0, sizeof p);
      w.i32(engine_->policy_get(g, &p));
      w.pod(p);
      return w.data();
    }
    case OP_POLICY_REGISTER: {
      uint32_t g = r.u32(), cond = r.u32(), client_reg = r.u32();
      std::weak_ptr<Client> wc = c;
      gm_policy_reg_t reg = 0;
      gm_status_t st = engine_->policy_register(
          g, cond,
          [wc, client_reg](const gm_violation_t& v) {
            auto cl = wc.lock();
            if (!cl || !cl->alive) return;
            WireWriter n;
            n.pod(v);
            std::lock_guard<std::mutex> l(cl->write_mu);
            if (cl->fd >= 0) rpc_send_frame(cl->fd, OP_NOTIFY_VIOLATION, client_reg, n.data());
          },
          &reg);
      if (st == GM_ST_OK) c->regs.insert(reg);
      w.i32(st);
      w.u32(reg);
      return w.data();
    }
    case OP_POLICY_UNREGISTER: {
      uint32_t reg = r.u32();
      c->regs.erase(reg);
      return status_only(engine_->policy_unregister(reg));
    }
    case OP_WATCH_PID_FIELDS: {
      uint32_t g = r.u32();
      int64_t freq = r.i64();
      double age = r.f64();
      int32_t keep = r.i32();
      if (!r.ok()) return status_only(GM_ST_BADPARAM);
      return status_only(engine_->watch_pid_fields(g, freq, age, keep));
    }
    case OP_PID_INFO: {
      uint32_t g = r.u32(), pid = r.u32();
      std::unique_ptr<gm_pid_info_t> info(new gm_pid_info_t);
      memset(info.get(), 0, sizeof *info);  // num_gpus must be 0, not heap garbage, when the engine returns early
      info->version = GM_PID_INFO_VERSION;
      gm_status_t st = engine_->pid_info(g, pid, info.get());
      w.i32(st);
      // Only the populated prefix travels.
      w.u32(info->num_gpus);
      w.u32((uint32_t)sizeof(gm_pid_gpu_stats_t));
      if (info->num_gpus) w.raw(info->gpus, info->num_gpus * sizeof(gm_pid_gpu_stats_t));
      return w.data();
    }
    case OP_INTROSPECT_TOGGLE: return status_only(engine_->introspect_toggle(r.u32() != 0));
    case OP_INTROSPECT: {
      gm_introspect_t in;
      memset(&in, 0, sizeof in);  // an error return leaves it untouched: never send stack bytes
      in.version = GM_INTROSPECT_VERSION;
      w.i32(engine_->introspect(&in));
      w.pod(in);
      return w.data();
    }
    case OP_EVENT_SUBSCRIBE: {
      uint32_t client_sub = r.u32();
      std::weak_ptr<Client> wc = c;
      uint32_t sub = 0;
      gm_status_t st = engine_->event_subscribe(
          [wc, client_sub](const gm_event_t& e) {
            auto cl = wc.lock();
            if (!cl || !cl->alive) return;
            WireWriter n;
            n.pod(e);
            std::lock_guard<std::mutex> l(cl->write_mu);
            if (cl->fd >= 0) rpc_send_frame(cl->fd, OP_NOTIFY_EVENT, client_sub, n.data());
          },
          &sub);
      if (st == GM_ST_OK) c->subs.insert(sub);
      w.i32(st);
      w.u32(sub);
      return w.data();
    }
    case OP_EVENT_UNSUBSCRIBE: {
      uint32_t sub = r.u32();
      c->subs.erase(sub);
      return status_only(engine_->event_unsubscribe(sub));
    }
  }
  return status_only(GM_ST_NOT_SUPPORTED);
}

}  // namespace gpumon
