// gpumon-exporter — the node metrics exporter.  Flag-compatible with the
// reference's script (`-h -e -p -o FILE -d MS`, NODE_NAME GPU selection:
// exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:5-78) and extended with
// a built-in HTTP endpoint, pod attribution and one-shot mode.
//
//   -e               run the host engine inside this process (embedded)
//   -p               also export the five profiling metrics
//   --prof-counters[=child]  with -e: feed dcgm_fi_prof_pipe_tensor_active, dcgm_fi_prof_sm_active and
//                    gpumon_prof_mem_pipe_active from hardware counters; implies -p.  Plain: libgpumon_prof.so inside
//                    this process (rocprofiler-sdk device counting + HSA; continuous windows; = GPUMON_PROF_COUNTERS=1).
//                    =child: a short-lived `gpumon-prof-probe --once` every --prof-period seconds takes the windows, so
//                    this process never opens the compute runtime (sampled values; = GPUMON_PROF_COUNTERS=child)
//   --prof-period S  seconds between probe runs with --prof-counters=child (default 30: one probe run costs about 0.3 s of CPU; = GPUMON_PROF_PERIOD_S)
//   -o FILE          publish atomically to FILE (default /run/prometheus/dcgm.prom; "" disables)
//   -d MS            collection interval, >= 100 (default 1000)
//   --slow-interval MS   period of the sources that cost a device-library call each and change slowly (error
//                    counters, bad pages, performance level / power cap, memory usage) when -d is shorter (default 1000)
//   --listen ADDR    serve /metrics, /gpu/metrics, /healthz (e.g. :9400)
//   --rest           also serve the REST routes on the listener
//   --gpus LIST      GPU indices to export ("-1" = all); NODE_NAME env lookup otherwise
//   --extras         AMD-only families (per-link xGMI, per-stack HBM, per-XCD, RAS blocks)
//   --self-metrics   exporter / engine self metrics
//   --kubelet-socket PATH   enable pod attribution through the PodResources API
//   --pod-map FILE   pod map fallback (TSV or kubelet device-plugin checkpoint)
//   --pod-refresh MS period of the pod-map query, on its own thread and one kept kubelet connection (default 10000)
//   --resource NAMES device-plugin resource names (default amd.com/gpu)
//   --diag-results PATH   export the report(s) left by `gpumon-diag --publish` as gpumon_diag_* metrics
//   --state-file PATH     keep the gpu_last_not_idle_time latch across restarts
//   --health         health check of every exported GPU on each sweep: gpumon_health_status{system=...},
//                    gpumon_health_overall, gpumon_health_incidents_total{system=...,level=...}
//   --legacy-layout  GPU-major output exactly like the reference's awk program
//   --once           one sweep, then exit
//   -connect ADDR [-socket]   use a running gpumon-hostengine instead of -e
#include <unistd.h>

#include "cli_common.h"

static void usage(const char* argv0) {
  fprintf(stderr, "usage: %s [-h] [-e] [-p] [-o output_file] [-d collect_interval_ms (>=100)] [--listen addr] [--once]\n", argv0);
}

// NODE_NAME=<node> plus an env var named after the node (dashes -> underscores)
// holding a comma list of GPU indices selects the GPUs to export
// (dcgm-exporter:52-62).
static std::string gpus_from_node_env() {
  const char* node = getenv("NODE_NAME");
  if (!node || !*node) return "-1";
  std::string var = node;
  for (auto& c : var)
    if (c == '-') c = '_';
  const char* v = getenv(var.c_str());
  fprintf(stderr, "Node: %s\n", var.c_str());
  return v && *v ? v : "-1";
}

int main(int argc, char** argv) {
  std::vector<std::string> args(argv + 1, argv + argc);
  CliConn conn;
  if (!parse_conn(args, &conn)) {
    usage(argv[0]);
    return 1;
  }
  bool embedded = false, profiling = false, once = false, extras = false, self = false, rest = false, legacy = false, health = false;
  std::string out = "/run/prometheus/dcgm.prom", listen, gpus, kubelet, podmap, resource, diagres, statefile;
  long interval = 1000, slow_interval = 0, pod_refresh = 0;
  for (size_t i = 0; i < args.size(); i++) {
    const std::string& a = args[i];
    auto val = [&](std::string* dst) {
      if (i + 1 >= args.size()) {
        usage(argv[0]);
        exit(1);
      }
      *dst = args[++i];
    };
    std::string tmp;
    if (a == "-h" || a == "--help") { usage(argv[0]); return 1; }
    else if (a == "-e") embedded = true;
    else if (a == "-p") profiling = true;
    else if (a == "--prof-counters") { profiling = true; setenv("GPUMON_PROF_COUNTERS", "1", 1); }
    else if (a == "--prof-counters=child") { profiling = true; setenv("GPUMON_PROF_COUNTERS", "child", 1); }
    else if (a == "--prof-period") { val(&tmp); setenv("GPUMON_PROF_PERIOD_S", tmp.c_str(), 1); }
    else if (a == "-o") val(&out);
    else if (a == "-d") { val(&tmp); interval = atol(tmp.c_str()); }
    else if (a == "--slow-interval") { val(&tmp); slow_interval = atol(tmp.c_str()); }
    else if (a == "--pod-refresh") { val(&tmp); pod_refresh = atol(tmp.c_str()); }
    else if (a == "--listen") val(&listen);
    else if (a == "--rest") rest = true;
    else if (a == "--gpus") val(&gpus);
    else if (a == "--extras") extras = true;
    else if (a == "--self-metrics") self = true;
    else if (a == "--kubelet-socket") val(&kubelet);
    else if (a == "--pod-map") val(&podmap);
    else if (a == "--resource") val(&resource);
    else if (a == "--diag-results") val(&diagres);
    else if (a == "--state-file") val(&statefile);
    else if (a == "--legacy-layout") legacy = true;
    else if (a == "--health") health = true;
    else if (a == "--once") once = true;
    else { usage(argv[0]); return 1; }
  }
  if (interval < 100) {  // the reference refuses intervals under 100 ms
    usage(argv[0]);
    return 1;
  }
  if (gpus.empty()) gpus = gpus_from_node_env();
  install_signals();

  // Readiness gate: keep trying until the device library answers, the way the
  // reference loops on `nvidia-smi` (dcgm-exporter:45-50).
  gm_status_t st;
  bool explicit_conn = !conn.connect.empty() || conn.start_hostengine;
  if (!embedded && !explicit_conn) {
    // Neither -e nor -connect: a host engine on the default port if there is one, else embedded.
    CliConn probe = conn;
    probe.connect = "127.0.0.1:5555";
    st = cli_init(probe);
    if (st != GM_ST_OK) {
      fprintf(stderr, "No host engine on %s; running the engine in-process.\n", probe.connect.c_str());
      embedded = true;
    }
  } else {
    st = GM_ST_UNINITIALIZED;
  }
  if (embedded || explicit_conn) {
    if (embedded && !explicit_conn) fprintf(stderr, "Starting the embedded host engine...\n");
    while ((st = cli_init(conn)) != GM_ST_OK) {
      if (once || g_stop) {
        fprintf(stderr, "error: cannot initialise the engine: %s\n", gm_strerror(st));
        return 1;
      }
      fprintf(stderr, "engine not ready (%s); retrying in 2 s\n", gm_strerror(st));
      for (int i = 0; i < 40 && !g_stop; i++) usleep(50000);
    }
  }

  gm_exporter_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.version = GM_EXPORTER_CONFIG_VERSION;
  cfg.profiling = profiling;
  cfg.amd_extras = extras;
  cfg.self_metrics = self;
  cfg.rest_api = rest;
  cfg.legacy_layout = legacy;
  cfg.health = health;
  cfg.interval_ms = interval;
  cfg.slow_interval_ms = (int32_t)slow_interval;
  cfg.pod_refresh_ms = (int32_t)pod_refresh;
  cfg.gpus = gpus.c_str();
  cfg.output_path = out.empty() ? nullptr : out.c_str();
  cfg.listen = listen.empty() ? nullptr : listen.c_str();
  cfg.kubelet_socket = kubelet.empty() ? nullptr : kubelet.c_str();
  cfg.pod_map_path = podmap.empty() ? nullptr : podmap.c_str();
  cfg.diag_results_path = diagres.empty() ? nullptr : diagres.c_str();
  cfg.state_path = statefile.empty() ? nullptr : statefile.c_str();
  cfg.resource_names = resource.empty() ? nullptr : resource.c_str();
  gm_exporter_t ex;
  CLI_CHECK(gm_exporter_create(&cfg, &ex));
  fprintf(stderr, "Collecting metrics at %s every %ldms from GPUs:%s (-1 means all GPUs).\n", out.empty() ? "(no file)" : out.c_str(),
          interval, gpus.c_str());
  if (profiling) fprintf(stderr, "Collecting profiling metrics as well.\n");
  if (once) {
    CLI_CHECK(gm_exporter_sweep(ex));
    if (out.empty()) {
      std::vector<char> buf(1 << 20);
      size_t len = buf.size();
      if (gm_exporter_render(ex, buf.data(), &len) == GM_ST_OK) fputs(buf.data(), stdout);
    }
  } else {
    CLI_CHECK(gm_exporter_start(ex));
    if (!listen.empty()) {
      int port = 0;
      gm_exporter_http_port(ex, &port);
      fprintf(stderr, "Serving /metrics on port %d.\n", port);
    }
    while (!g_stop) usleep(100000);
    fprintf(stderr, "Caught signal, terminating...\n");
    gm_exporter_stop(ex);
  }
  gm_exporter_destroy(ex);
  gm_shutdown();
  fprintf(stderr, "Done\n");
  return 0;
}
