#pragma once

#include <atomic>
#include <string>
#include <thread>

#include "exporter.h"
#include "http.h"

namespace gpumon {

// Defaults are the compile-time constants of the reference
// (src/watchers.go:15-21, main.go:11, kubelet_server.go:15-18, device_pod.go:17),
// all of them configurable here.
struct PodExporterConfig {
  std::string source_path = "/run/prometheus/dcgm.prom";
  std::string dest_path = "/run/dcgm/dcgm-pod.prom";
  std::string kubelet_socket = "/var/lib/kubelet/pod-resources/kubelet.sock";
  std::string resource_names = "amd.com/gpu";
  std::string index_prefix = "amdgpu";
  std::string listen = ":9400";
  int timeout_ms = 10000;
  int liveness_ms = 600000;
  int pod_refresh_ms = 0;  // 0 = query the kubelet on every sweep, like the reference
};

class PodExporter {
 public:
  explicit PodExporter(const PodExporterConfig& cfg);
  ~PodExporter();
  bool start(std::string* error);
  void stop();
  bool rewrite_once(std::string* error);
  bool dead() const { return dead_.load(); }
  int http_port() const { return http_.port(); }
  uint64_t rewrites() const { return rewrites_.load(); }

 private:
  void loop(const std::string& base);
  PodExporterConfig cfg_;
  std::atomic<bool> running_{false}, dead_{false};
  std::atomic<uint64_t> rewrites_{0};
  int ino_fd_ = -1;
  std::thread th_;
  HttpServer http_;
  PodMap map_;
  bool have_map_ = false;
  int64_t last_map_us_ = 0;
  int kubelet_failures_ = 0;
};

}  // namespace gpumon
