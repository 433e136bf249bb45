// Native Prometheus exporter: sampler cache -> text renderer -> atomic file
// sink and/or in-process HTTP.  One process, no forks, no text re-parsing; it
// replaces the `dcgmi dmon | gawk` pipeline of the reference
// (exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:85-194) and folds in
// the pod-attribution rewrite of its companion exporter
// (exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src/device_pod.go:57-113).
#pragma once

#include <array>
#include <atomic>
#include <condition_variable>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../engine/engine.h"
#include "http.h"

namespace gpumon {

struct PodInfo {
  std::string pod, ns, container;
};
using PodMap = std::map<std::string, PodInfo>;  // device id -> pod

// TSV: "<device-id>\t<pod>\t<namespace>\t<container>\n".
PodMap parse_pod_map_tsv(const std::string& tsv);
std::string pod_map_to_tsv(const PodMap& m);
// Kubelet device-plugin checkpoint (JSON) fallback: device id -> pod UID / container.
bool parse_device_plugin_checkpoint(const std::string& json, const std::string& resource_names, PodMap* out);
// Appends pod labels to the samples of an existing exposition text.
std::string relabel_text(const std::string& text, const PodMap& m, const std::string& index_prefix);
// Kubelet PodResources List() over a unix socket (gRPC, hand-rolled HTTP/2).
gm_status_t kubelet_list_pod_resources(const std::string& socket_path, const std::string& resource_names,
                                       int timeout_ms, PodMap* out, std::string* error);

// The same call on a connection that is kept between calls (the reference dials the kubelet anew once a second:
// src/watchers.go:41 -> kubelet_server.go:36-41).  A connection that went away (kubelet restart, GOAWAY) is replaced
// once per call.  Not thread safe: one owner.
class KubeletClient {
 public:
  KubeletClient(const std::string& socket_path, const std::string& resource_names);
  ~KubeletClient();
  KubeletClient(const KubeletClient&) = delete;
  KubeletClient& operator=(const KubeletClient&) = delete;
  gm_status_t list(int timeout_ms, PodMap* out, std::string* error);
  uint64_t connects() const { return connects_; }  // connections opened so far

 private:
  struct Impl;
  void disconnect();
  gm_status_t connect_now(std::string* error);
  gm_status_t list_once(PodMap* out, std::string* error, bool* transport);
  std::string socket_path_, resource_names_;
  Impl* impl_;
  uint64_t connects_ = 0;
};

struct ExporterConfig {
  bool profiling = false;
  bool amd_extras = false;
  bool self_metrics = false;
  bool rest_api = false;
  bool legacy_layout = false;
  bool health = false;  // health check of every exported GPU once per sweep, exported as gpumon_health_*
  int64_t interval_ms = 1000;
  // Sources that cost a device-library call of their own and change slowly (error counters, bad pages, performance
  // level and power cap, memory usage) are sampled at this period when the collection interval is shorter.
  int64_t slow_interval_ms = 1000;
  int64_t pod_refresh_ms = 10000;  // how often the pod map is asked for (kubelet or file), on a thread of its own
  std::string gpus;  // "" / "-1" = all
  std::string output_path;
  std::string listen;
  std::string kubelet_socket;
  std::string resource_names = "amd.com/gpu";
  std::string pod_map_path;
  std::string diag_results_path;  // report(s) of gpumon-diag --publish, rendered as gpumon_diag_* families
  std::string state_path;         // carries the last-not-idle latch across restarts ("" = in memory only)
};

class Exporter {
 public:
  Exporter(Engine* engine, const ExporterConfig& cfg);
  ~Exporter();
  gm_status_t init();
  void render(std::string* out);  // from the cache; never samples
  void update_latch();            // the last-not-idle latch alone: what a sweep that writes no file still owes
  gm_status_t publish();          // render + publish file
  gm_status_t sweep();            // force-sample, then publish
  gm_status_t start();
  void stop();
  int http_port() const { return http_.port(); }
  void set_pod_map(const PodMap& m);
  const std::vector<uint32_t>& gpus() const { return gpus_; }

 private:
  void loop();
  void pod_loop();
  void refresh_pod_map();
  void handle_http(const HttpRequest& req, HttpResponse* resp);
  void render_self(std::string* out);
  void check_health();
  void render_health(const std::vector<std::string>& base, const std::vector<std::string>& podl, std::string* out);
  void load_state();
  void save_state(bool force);

  Engine* engine_;
  ExporterConfig cfg_;
  std::vector<uint32_t> gpus_;
  std::vector<std::string> uuids_, bdfs_;
  std::vector<int> partitions_;  // compute-partition index per exported entry; -1 = GPU is not partitioned (no label)
  std::vector<std::vector<int>> link_peer_;  // per exported entry: GPU index at the far end of xGMI link l, -1 = unknown
  std::vector<uint16_t> fields_;  // every field id fetched per GPU
  gm_group_t group_ = 0;
  gm_field_group_t fg_ = 0, fg_slow_ = 0;  // per-tick fields; fields of the slow sources (0 = none split off)
  bool watching_ = false;
  std::unique_ptr<KubeletClient> kubelet_;  // owned by whoever refreshes the pod map: init(), then the pod thread
  std::thread pod_th_;

  std::mutex mu_;  // last_not_idle_, pod_map_, cached text
  std::map<uint32_t, int64_t> last_not_idle_;
  std::vector<std::string> proc_source_;  // per exported GPU, refreshed once a minute (render_self)
  int64_t proc_source_mono_ = 0;
  int64_t diag_polled_mono_ = 0;   // publish thread only
  bool kubelet_failing_ = false;   // pod thread only
  int64_t kubelet_warned_mono_ = 0;
  bool state_dirty_ = false;
  int64_t state_saved_mono_ = 0;
  PodMap pod_map_;
  // Health of every exported GPU as of the last sweep (under mu_): level per subsystem bit, and how many warnings and
  // failures each subsystem has reported since the exporter started (an incident that lasts one check is still seen
  // by a scraper that looks at the counter's increase).
  struct HealthCell {
    int32_t level = 0;
    uint64_t warnings = 0, failures = 0;
  };
  static constexpr int kHealthSystems = 8;
  std::vector<std::array<HealthCell, kHealthSystems>> health_;
  std::vector<int32_t> health_overall_;
  bool health_valid_ = false;
  // Policy violations of the exported GPUs since the exporter started, per condition bit (default thresholds, or the
  // ones set on the engine); counted on the engine's notifier thread.
  static constexpr int kPolicyConditions = 7;
  std::mutex violation_mu_;
  std::vector<std::array<uint64_t, kPolicyConditions>> violations_;
  gm_policy_reg_t policy_reg_ = 0;
  bool policy_registered_ = false;
  std::string counter_note_;  // appended to the HELP of the counter-fed families when their values are samples
  std::string cached_;  // last published text (served over HTTP between sweeps)
  std::vector<gm_value_t> scratch_;

  std::atomic<bool> running_{false};
  std::thread th_;
  std::mutex wake_mu_;
  std::condition_variable wake_cv_;
  HttpServer http_;

  std::atomic<uint64_t> sweeps_{0}, scrapes_{0}, publish_errors_{0};
  std::atomic<int64_t> last_render_us_{0}, last_sweep_us_{0}, last_pod_refresh_mono_{0};
  std::atomic<int> pod_map_ok_{-1};
  std::atomic<uint64_t> pod_refreshes_{0}, kubelet_connects_{0};
};

// REST routes of the reference's sample server (bindings/go/samples/dcgm/restApi/
// server.go:40-71) under /dcgm/... with /gpumon/... aliases.  Returns false
// when the path is not a REST route.
// Appends the gpumon_diag_* families for the report file (or directory of *.json reports) at `path`.
// Also hands the verdicts of every report that is new or has changed to `engine` (may be null), where a failed
// test becomes a health incident and a GPU event.
void render_diag_results(const std::string& path, std::string* out, Engine* engine = nullptr);
bool handle_rest(Engine* engine, const HttpRequest& req, HttpResponse* resp);

}  // namespace gpumon
