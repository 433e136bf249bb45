// File-watch pod exporter: drop-in for the reference's pod-gpu-metrics-exporter
// (exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src/{main,watchers,
// device_pod,file_utils,http}.go).  Watches the directory a node exporter
// publishes into, and every time the metrics file is (re)created it rewrites a
// copy with pod_name / pod_namespace / container_name labels and serves that
// copy over HTTP at /gpu/metrics.
//
// Differences kept on purpose: the kubelet connection is not re-dialled on every
// sweep when a refresh interval is configured, rename() replaces fork+mv, and a
// kubelet outage serves the last good pod map instead of dropping attribution.
#include <cerrno>
#include <cstring>
#include <poll.h>
#include <sys/inotify.h>
#include <sys/stat.h>
#include <unistd.h>

#include "../common/util.h"
#include "exporter.h"
#include "pod_exporter.h"

namespace gpumon {

PodExporter::PodExporter(const PodExporterConfig& cfg) : cfg_(cfg) {}
PodExporter::~PodExporter() { stop(); }

bool PodExporter::rewrite_once(std::string* error) {
  std::string text;
  if (!read_file(cfg_.source_path, &text)) {
    if (error) *error = "failed to open " + cfg_.source_path + ": " + strerror(errno);
    return false;
  }
  int64_t now = monotonic_us();
  if (!have_map_ || now - last_map_us_ >= (int64_t)cfg_.pod_refresh_ms * 1000) {
    PodMap m;
    std::string err;
    gm_status_t st = kubelet_list_pod_resources(cfg_.kubelet_socket, cfg_.resource_names, cfg_.timeout_ms, &m, &err);
    if (st == GM_ST_OK) {
      map_ = m;
      have_map_ = true;
      last_map_us_ = now;
      kubelet_failures_ = 0;
    } else {
      // the first failure of a row, then every sixtieth: one line a minute at the default one-second sweep
      if (kubelet_failures_++ % 60 == 0) GM_LOGW("failed to get devices Pod information: %s", err.c_str());
      if (!have_map_) {
        if (error) *error = err;
        return false;
      }
    }
  }
  std::string outp = relabel_text(text, map_, cfg_.index_prefix);
  size_t slash = cfg_.dest_path.rfind('/');
  if (slash != std::string::npos && slash > 0) mkdir(cfg_.dest_path.substr(0, slash).c_str(), 0755);
  if (!write_file_atomic(cfg_.dest_path, outp.data(), outp.size(), 0644)) {
    if (error) *error = "error writing " + cfg_.dest_path + ": " + strerror(errno);
    return false;
  }
  rewrites_++;
  return true;
}

bool PodExporter::start(std::string* error) {
  if (running_.exchange(true)) return true;
  size_t slash = cfg_.source_path.rfind('/');
  std::string dir = slash == std::string::npos ? "." : cfg_.source_path.substr(0, slash);
  std::string base = slash == std::string::npos ? cfg_.source_path : cfg_.source_path.substr(slash + 1);
  mkdir(dir.c_str(), 0755);
  ino_fd_ = inotify_init1(IN_NONBLOCK | IN_CLOEXEC);
  if (ino_fd_ < 0 || inotify_add_watch(ino_fd_, dir.c_str(), IN_CREATE | IN_MOVED_TO | IN_CLOSE_WRITE) < 0) {
    if (error) *error = "cannot watch " + dir + ": " + strerror(errno);
    running_ = false;
    if (ino_fd_ >= 0) close(ino_fd_);
    ino_fd_ = -1;
    return false;
  }
  if (!cfg_.listen.empty()) {
    bool ok = http_.start(cfg_.listen, [this](const HttpRequest& rq, HttpResponse* rs) {
      if (rq.path == "/gpu/metrics" || rq.path == "/metrics") {
        if (!read_file(cfg_.dest_path, &rs->body)) {
          rs->status = 500;
          rs->body = "metrics file not available yet\n";
        }
        rs->content_type = "text/plain; version=0.0.4; charset=utf-8";
      } else if (rq.path == "/healthz") {
        rs->body = "ok\n";
      } else {
        rs->status = 404;
        rs->body = "404 page not found\n";
      }
    }, cfg_.timeout_ms);
    if (!ok) {
      if (error) *error = "cannot listen on " + cfg_.listen;
      running_ = false;
      close(ino_fd_);
      ino_fd_ = -1;
      return false;
    }
  }
  th_ = std::thread([this, base] { loop(base); });
  return true;
}

void PodExporter::loop(const std::string& base) {
  int64_t last_event = monotonic_us();
  // A file that already exists is handled right away.
  struct stat sb;
  if (stat(cfg_.source_path.c_str(), &sb) == 0) {
    std::string err;
    if (!rewrite_once(&err)) GM_LOGW("%s", err.c_str());
  }
  alignas(struct inotify_event) char buf[4096];
  while (running_) {
    pollfd p = {ino_fd_, POLLIN, 0};
    int rc = poll(&p, 1, 250);
    if (rc > 0) {
      ssize_t n = read(ino_fd_, buf, sizeof buf);
      bool hit = false;
      for (char* q = buf; n > 0 && q < buf + n;) {
        struct inotify_event* ev = (struct inotify_event*)q;
        // `mv tmp file` shows up as MOVED_TO, a plain writer as CREATE / CLOSE_WRITE.
        if (ev->len && base == ev->name && (ev->mask & (IN_CREATE | IN_MOVED_TO | IN_CLOSE_WRITE))) hit = true;
        q += sizeof(struct inotify_event) + ev->len;
      }
      if (hit) {
        last_event = monotonic_us();
        GM_LOGD("inotify: %s created, now adding device pod information", cfg_.source_path.c_str());
        std::string err;
        if (!rewrite_once(&err)) GM_LOGE("%s", err.c_str());
      }
    }
    // No sweeps for the liveness window: report dead so the pod gets restarted
    // (the reference calls glog.Fatal after 10 minutes: src/watchers.go:56-59).
    if (cfg_.liveness_ms > 0 && monotonic_us() - last_event > (int64_t)cfg_.liveness_ms * 1000) {
      GM_LOGE("no metrics file events for %d ms; giving up", cfg_.liveness_ms);
      dead_ = true;
      break;
    }
  }
}

void PodExporter::stop() {
  if (!running_.exchange(false)) return;
  if (th_.joinable()) th_.join();
  http_.stop();
  if (ino_fd_ >= 0) close(ino_fd_);
  ino_fd_ = -1;
}

}  // namespace gpumon
