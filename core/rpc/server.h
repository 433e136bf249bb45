// RPC server: exposes a LocalEngine to remote libgpumon clients (the
// nv-hostengine role).  One thread per connection; everything a connection
// created (groups, field groups, policy registrations, event subscriptions)
// is released when it disconnects.
#pragma once

#include <atomic>
#include <memory>
#include <mutex>
#include <set>
#include <string>
#include <thread>
#include <vector>

#include "../engine/engine.h"

namespace gpumon {

class RpcServer {
 public:
  explicit RpcServer(Engine* engine);
  ~RpcServer();
  gm_status_t start(const std::string& address, bool is_unix);
  void stop();
  const std::string& address() const { return address_; }

 private:
  struct Client {
    int fd = -1;
    std::mutex write_mu;
    std::atomic<bool> alive{true};
    std::set<gm_group_t> groups;
    std::set<gm_field_group_t> field_groups;
    std::set<gm_policy_reg_t> regs;
    std::set<uint32_t> subs;
    std::thread th;
  };
  void accept_loop();
  void serve(std::shared_ptr<Client> c);
  std::string dispatch(const std::shared_ptr<Client>& c, uint32_t op, const std::string& req);

  Engine* engine_;
  int listen_fd_ = -1;
  bool is_unix_ = false;
  std::string address_;
  std::atomic<bool> running_{false};
  std::thread accept_th_;
  std::mutex mu_;
  std::vector<std::shared_ptr<Client>> clients_;
};

// Client side.
std::unique_ptr<Engine> make_remote_engine(const std::string& address, bool is_unix, int timeout_ms,
                                           gm_status_t* status, std::string* error);

}  // namespace gpumon
