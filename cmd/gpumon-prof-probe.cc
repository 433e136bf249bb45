// gpumon-prof-probe: prints what the profiling-counter plugin sees, once per interval.
//   gpumon-prof-probe [--interval-ms N] [--count N] [--lib PATH]
//   gpumon-prof-probe --once [--interval-ms N]
// One line per GPU agent and window: percent of wall time the graphics engine, the shader arrays, the
// memory pipeline and the MFMA pipes were busy (each window measures the matrix pipe and ONE of the
// others: the GRBM block has two counter slots), the shader clock held, and the folded hardware counters
// behind them ($GPUMON_PROF_GRBM=A,B,.. picks the rotating GRBM counters, $GPUMON_PROF_EXTRA=A,B,C adds
// up to three SQ counters to look at).
// The counters are device-wide: run a workload in another process and watch it here.
//
// --once runs one window per counter set and prints ONE line per agent,
//   sample pci=0000:5d:00.0 window_s=0.750 gpu_active=99.1 cu_active=98.7 mem_pipe_active=61.0 mfma_active=84.2 clock_mhz=1934
// ("nan" for what the device cannot give), then exits: the form a monitoring daemon starts as a short-lived
// child every few seconds, so that it does not keep the compute runtime open itself.
#include <dlfcn.h>
#include <sys/prctl.h>
#include <signal.h>
#include <sys/resource.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <string>
#include <vector>

#include "../core/prof/prof_counters.h"

int main(int argc, char** argv) {
  prctl(PR_SET_PDEATHSIG, SIGTERM);  // started as a child by gpumon-diag: never outlive the run it watches
  int interval_ms = 1000, count = 5;
  bool once = false;
  std::string lib;
  for (int i = 1; i < argc; i++) {
    if (!strcmp(argv[i], "--interval-ms") && i + 1 < argc) interval_ms = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--count") && i + 1 < argc) count = atoi(argv[++i]);
    else if (!strcmp(argv[i], "--lib") && i + 1 < argc) lib = argv[++i];
    else if (!strcmp(argv[i], "--once")) once = true;
    else {
      fprintf(stderr, "usage: %s [--interval-ms N] [--count N] [--once] [--lib PATH]\n", argv[0]);
      return 2;
    }
  }
  if (lib.empty()) {  // next to this binary: build/bin/.. = build/
    char self[4096];
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    std::string dir = n > 0 ? std::string(self, (size_t)n) : std::string(".");
    dir = dir.substr(0, dir.rfind('/'));
    lib = dir + "/../libgpumon_prof.so";
  }
  void* h = dlopen(lib.c_str(), RTLD_NOW | RTLD_LOCAL);
  if (!h) {
    fprintf(stderr, "dlopen %s: %s\n", lib.c_str(), dlerror());
    return 1;
  }
  auto open_fn = (decltype(&gmp_open))dlsym(h, "gmp_open");
  auto info_fn = (decltype(&gmp_agent_info))dlsym(h, "gmp_agent_info");
  auto sample_fn = (decltype(&gmp_sample))dlsym(h, "gmp_sample");
  auto err_fn = (decltype(&gmp_last_error))dlsym(h, "gmp_last_error");
  auto close_fn = (decltype(&gmp_close))dlsym(h, "gmp_close");
  auto abi_fn = (decltype(&gmp_abi_version))dlsym(h, "gmp_abi_version");
  auto sets_fn = (decltype(&gmp_set_count))dlsym(h, "gmp_set_count");
  if (!open_fn || !info_fn || !sample_fn || !err_fn || !close_fn || !abi_fn || !sets_fn || abi_fn() != GMP_ABI_VERSION) {
    fprintf(stderr, "%s: missing symbols or another version of the sample layout\n", lib.c_str());
    return 1;
  }
  char err[512] = "";
  const int n = open_fn(err, sizeof err);
  if (n < 0) {
    fprintf(stderr, "gmp_open: %s\n", err);
    return 1;
  }
  if (once) {
    if (interval_ms == 1000) interval_ms = 250;
    const int sets = sets_fn() > 0 ? sets_fn() : 1;
    struct Acc {
      double gpu = NAN, cu = NAN, mem = NAN, clock = NAN, mfma_sum = 0, window = 0;
      int mfma_n = 0;
    };
    std::vector<Acc> acc((size_t)n);
    int rc = 0;
    for (int i = 0; i <= sets; i++) {
      for (int a = 0; a < n; a++) {
        gmp_sample_t s;
        const int r = sample_fn(a, &s);
        if (r < 0) {
          fprintf(stderr, "agent %d: %s\n", a, err_fn());
          rc = 1;
        } else if (r == 0) {
          Acc& x = acc[(size_t)a];
          if (!std::isnan(s.gpu_active)) x.gpu = s.gpu_active;
          if (!std::isnan(s.cu_active)) x.cu = s.cu_active;
          if (!std::isnan(s.mem_pipe_active)) x.mem = s.mem_pipe_active;
          if (!std::isnan(s.clock_mhz)) x.clock = s.clock_mhz;
          if (!std::isnan(s.mfma_active)) x.mfma_sum += s.mfma_active, x.mfma_n++;
          x.window += s.window_s;
        }
      }
      if (i < sets) usleep((useconds_t)interval_ms * 1000);
    }
    for (int a = 0; a < n; a++) {
      gmp_agent_info_t ai;
      if (info_fn(a, &ai) != 0) continue;
      const Acc& x = acc[(size_t)a];
      printf("sample pci=%04x:%02x:%02x.%x window_s=%.3f gpu_active=%.2f cu_active=%.2f mem_pipe_active=%.2f mfma_active=%.2f clock_mhz=%.0f\n",
             ai.pci_domain, (ai.pci_location >> 8) & 0xff, (ai.pci_location >> 3) & 0x1f, ai.pci_location & 7, x.window, x.gpu, x.cu,
             x.mem, x.mfma_n ? x.mfma_sum / x.mfma_n : NAN, x.clock);
    }
    fflush(stdout);
    close_fn();
    return rc;
  }
  printf("agents: %d\n", n);
  for (int a = 0; a < n; a++) {
    gmp_agent_info_t ai;
    if (info_fn(a, &ai) == 0)
      printf("agent %d: %s  pci %04x:%02x:%02x.%x  cus %u  simds %u\n", a, ai.name, ai.pci_domain, (ai.pci_location >> 8) & 0xff,
             (ai.pci_location >> 3) & 0x1f, ai.pci_location & 7, ai.cu_count, ai.simd_count);
  }
  fflush(stdout);
  int rc = 0;
  auto cpu_s = []() {
    struct rusage ru;
    getrusage(RUSAGE_SELF, &ru);
    return ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec) * 1e-6;
  };
  const double cpu_open = cpu_s();
  double cpu_started = cpu_open;
  struct timespec t0;
  clock_gettime(CLOCK_MONOTONIC, &t0);
  for (int i = 0; i <= count; i++) {
    for (int a = 0; a < n; a++) {
      gmp_sample_t s;
      struct timespec c0, c1;
      clock_gettime(CLOCK_MONOTONIC, &c0);
      const double cpu0 = cpu_s();
      const int r = sample_fn(a, &s);
      clock_gettime(CLOCK_MONOTONIC, &c1);
      const double call_ms = (c1.tv_sec - c0.tv_sec) * 1e3 + (c1.tv_nsec - c0.tv_nsec) * 1e-6, call_cpu_ms = (cpu_s() - cpu0) * 1e3;
      if (r < 0) {
        fprintf(stderr, "agent %d: %s\n", a, err_fn());
        rc = 1;
      } else if (r == 0) {
        printf("agent %d window %.3fs %s %.2f (%d instances) mfma_active %.2f clock_mhz %.0f  raw count %.0f grbm %.0f mfma %.0f extra %.0f %.0f %.0f  (call %.1f ms wall, %.1f ms cpu)\n",
               a, s.window_s, s.grbm_name, s.grbm_busy, s.grbm_instances, s.mfma_active, s.clock_mhz, s.raw[0], s.raw[1], s.raw[2], s.raw[3], s.raw[4], s.raw[5], call_ms, call_cpu_ms);
      }
    }
    fflush(stdout);
    if (i == 0) {  // the first round only starts the counters: a one-time cost of its own
      cpu_started = cpu_s();
      clock_gettime(CLOCK_MONOTONIC, &t0);
    }
    if (i < count) usleep((useconds_t)interval_ms * 1000);
  }
  struct timespec t1;
  clock_gettime(CLOCK_MONOTONIC, &t1);
  const double wall = (t1.tv_sec - t0.tv_sec) + (t1.tv_nsec - t0.tv_nsec) * 1e-9;
  printf("cpu: %.3f s to open, %.3f s to start the counters, %.3f s over %.2f s of sampling (%.2f %% of one core)\n", cpu_open,
         cpu_started - cpu_open, cpu_s() - cpu_started, wall, wall > 0 ? 100.0 * (cpu_s() - cpu_started) / wall : 0.0);
  if (getenv("GPUMON_PROF_PROBE_THREADS")) {  // which thread the CPU time went to
    char path[64];
    snprintf(path, sizeof path, "/proc/%d/task", (int)getpid());
    std::string cmd = std::string("for t in ") + path + "/*; do echo \"$(cat $t/comm) $(cut -d' ' -f14,15 $t/stat 2>/dev/null || awk '{print $14, $15}' $t/stat)\"; done";
    fflush(stdout);
    if (system(cmd.c_str()) != 0) rc = rc;
  }
  close_fn();
  if (const char* e = getenv("GPUMON_PROF_PROBE_AFTER_CLOSE_S")) {  // does the plugin cost anything with its counters stopped?
    const double c0 = cpu_s();
    const int secs = atoi(e);
    sleep((unsigned)secs);
    printf("cpu with the counters stopped: %.3f s over %d s\n", cpu_s() - c0, secs);
  }
  return rc;
}
