// gpumon-pod-exporter — per-pod GPU metrics (file-watch mode).  Same job and
// defaults as the reference's pod-gpu-metrics-exporter: watch
// /run/prometheus/dcgm.prom, add pod labels from the kubelet PodResources API,
// publish /run/dcgm/dcgm-pod.prom and serve it at :9400/gpu/metrics
// (exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src/main.go:11-38).
// Every path, port and timeout is a flag instead of a compile-time constant.
#include <unistd.h>

#include "cli_common.h"

int main(int argc, char** argv) {
  std::vector<std::string> args(argv + 1, argv + argc);
  if (!expand_config(args)) return 2;  // --config FILE, $GPUMON_ARGS
  std::string src, dst, sock, res, prefix, listen = ":9400";
  int timeout = 0, liveness = 0, refresh = 0, verbosity = -1;
  for (size_t i = 0; i < args.size(); i++) {
    const std::string& a = args[i];
    auto val = [&](std::string* dst_) {
      if (i + 1 >= args.size()) {
        fprintf(stderr, "missing value for %s\n", a.c_str());
        exit(2);
      }
      *dst_ = args[++i];
    };
    std::string tmp;
    if (a == "--source") val(&src);
    else if (a == "--dest") val(&dst);
    else if (a == "--kubelet-socket") val(&sock);
    else if (a == "--resource") val(&res);
    else if (a == "--index-prefix") val(&prefix);
    else if (a == "--listen") val(&listen);
    else if (a == "--timeout-ms") { val(&tmp); timeout = atoi(tmp.c_str()); }
    else if (a == "--liveness-ms") { val(&tmp); liveness = atoi(tmp.c_str()); }
    else if (a == "--pod-refresh-ms") { val(&tmp); refresh = atoi(tmp.c_str()); }
    else if (a == "-v") { val(&tmp); verbosity = atoi(tmp.c_str()); }
    else if (a == "-logtostderr") {}  // accepted for drop-in compatibility with the glog flags
    else {
      fprintf(stderr,
              "usage: gpumon-pod-exporter [--source FILE] [--dest FILE] [--kubelet-socket PATH] [--resource NAMES]\n"
              "                           [--index-prefix P] [--listen ADDR] [--timeout-ms N] [--liveness-ms N]\n"
              "                           [--pod-refresh-ms N] [-v LEVEL]\n");
      return 2;
    }
  }
  if (verbosity >= 0) gm_set_log_level(verbosity >= 2 ? 3 : 2);
  install_signals();
  gm_pod_exporter_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.version = GM_POD_EXPORTER_CONFIG_VERSION;
  cfg.source_path = src.c_str();
  cfg.dest_path = dst.c_str();
  cfg.kubelet_socket = sock.c_str();
  cfg.resource_names = res.c_str();
  cfg.index_prefix = prefix.c_str();
  cfg.listen = listen.c_str();
  cfg.timeout_ms = timeout;
  cfg.liveness_ms = liveness;
  cfg.pod_refresh_ms = refresh;
  gm_pod_exporter_t pe;
  gm_status_t st = gm_pod_exporter_start(&cfg, &pe);
  if (st != GM_ST_OK) {
    fprintf(stderr, "error: cannot start the pod exporter: %s\n", gm_strerror(st));
    return 1;
  }
  fprintf(stderr, "Starting FS watcher.\nStarting HTTP server on %s\n", listen.c_str());
  int alive = 1;
  while (!g_stop && alive) {
    usleep(200000);
    gm_pod_exporter_status(pe, &alive, nullptr, nullptr);
  }
  gm_pod_exporter_stop(pe);
  if (!alive) {
    fprintf(stderr, "Something is too slow. Go fix it\n");
    return 1;
  }
  fprintf(stderr, "Exiting pod exporter\n");
  return 0;
}
