// gpumon-diag — active diagnostics for MI355X nodes (the `dcgmi diag` role; the
// reference itself ships no active test, only the passive health watches of
// bindings/go/dcgm/health.go:55-119).
//
//   gpumon-diag info                       HIP-side device facts
//   gpumon-diag hbm    [-g GPU] [--bytes N] [--iters N] [--kernel NAME|all] [--min-tbps X]
//   gpumon-diag memtest[-g GPU] [--bytes N | --fraction F] [--patterns a,b] [--passes N]
//   gpumon-diag mfma   [-g GPU] [--dtype bf16|fp8|mxfp8|f32|all] [--seconds S]
//   gpumon-diag gemm   [-g GPU] [--dtype bf16|fp8|mxfp8] [--m M --n N --k K] [--seconds S] [--no-telemetry] [--mfma-counters]
//   gpumon-diag rccl   [--gpus 0,1,..] [--ops a,b] [--min-bytes N] [--max-bytes N] [--scaling]
//   gpumon-diag links  [--gpus 0,1,..] [--bytes N] [--modes uni,bidi,all] [--floor-gbps X]   (alias: p2p)
//   gpumon-diag run -r 1|2|3 [-g GPU]      graded suites: quick / medium / long
// Every command takes --json and --publish FILE (leave the JSON report where
// `gpumon-exporter --diag-results` picks it up and exports it as gpumon_diag_* metrics).
// Exit status: 0 pass, 1 a test failed, 2 usage / setup error.
#include <unistd.h>
#include <signal.h>
#include <spawn.h>
#include <sys/wait.h>

#include <strings.h>

#include <algorithm>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <ctime>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>

#include "../diag/gpumon_diag.h"
#include "cli_common.h"

namespace {

struct Args {
  std::map<std::string, std::string> kv;
  bool has(const std::string& k) const { return kv.count(k) != 0; }
  std::string str(const std::string& k, const std::string& d = "") const {
    auto it = kv.find(k);
    return it == kv.end() ? d : it->second;
  }
  double num(const std::string& k, double d) const { return has(k) ? atof(str(k).c_str()) : d; }
  uint64_t size(const std::string& k, uint64_t d) const {
    if (!has(k)) return d;
    std::string s = str(k);
    double v = atof(s.c_str());
    char u = s.empty() ? 0 : (char)toupper(s.back());
    if (u == 'K') v *= 1ull << 10;
    else if (u == 'M') v *= 1ull << 20;
    else if (u == 'G') v *= 1ull << 30;
    else if (u == 'T') v *= 1ull << 40;
    return (uint64_t)v;
  }
};

const char* kFlagsWithValue[] = {"-g", "--bytes", "--iters", "--kernel", "--min-tbps", "--fraction", "--patterns", "--passes",
                                 "--dtype", "--seconds", "--m", "--n", "--k", "--gpus", "--ops", "--min-bytes",
                                 "--max-bytes", "-r", "--chunk", "--min-tflops", "--factor", "--publish", "--modes", "--floor-gbps",
                                 "--warn-ratio", "--fail-ratio", "--workgroups", "--submit", "--min-busbw"};

bool parse(const std::vector<std::string>& in, Args* a) {
  for (size_t i = 0; i < in.size(); i++) {
    bool takes = false;
    for (const char* f : kFlagsWithValue) takes = takes || in[i] == f;
    if (takes) {
      if (i + 1 >= in.size()) {
        fprintf(stderr, "missing value for %s\n", in[i].c_str());
        return false;
      }
      a->kv[in[i]] = in[i + 1];
      i++;
    } else if (in[i].rfind("-", 0) == 0) {
      a->kv[in[i]] = "1";
    } else {
      fprintf(stderr, "unexpected argument %s\n", in[i].c_str());
      return false;
    }
  }
  return true;
}

std::vector<std::string> csv(const std::string& s) {
  std::vector<std::string> out;
  size_t p = 0;
  while (p <= s.size()) {
    size_t c = s.find(',', p);
    if (c == std::string::npos) c = s.size();
    if (c > p) out.push_back(s.substr(p, c - p));
    p = c + 1;
  }
  return out;
}

// Collected output: a text report and a JSON document built side by side.
struct Report {
  bool json = false;
  bool failed = false;
  std::string js;  // comma-joined members of the top-level object
  void text(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    if (json) return;
    va_list ap;
    va_start(ap, fmt);
    vprintf(fmt, ap);
    va_end(ap);
    fflush(stdout);
  }
  void member(const std::string& key, const std::string& value) {
    if (!js.empty()) js += ",";
    js += "\"" + key + "\":" + value;
  }
  void fail(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    failed = true;
    va_list ap;
    va_start(ap, fmt);
    fprintf(stderr, "FAIL: ");
    vfprintf(stderr, fmt, ap);
    fprintf(stderr, "\n");
    va_end(ap);
  }
  // Per-test verdicts, the part of the report the host engine understands (gm_diag_result_t): a failure is kept
  // as the test's verdict and fails the run; a pass is recorded only where the test has no failure yet, so that a
  // later clean run replaces an old failure in the engine.
  struct Verdict {
    std::string test;
    uint32_t system;  // GM_HEALTH_WATCH_* bit
    int health;       // GM_HEALTH_PASS / WARN / FAIL
    double value;
    std::string message;
    int gpu = -1;  // HIP device the verdict is about; -1 = the report's GPU
  };
  std::vector<Verdict> verdicts;
  // A verdict about one GPU of a multi-GPU test (the link test): one per (test, gpu), the worst health wins.
  void verdict_for_gpu(int gpu, const std::string& test, uint32_t system, int health, double value, const std::string& message) {
    if (health == GM_HEALTH_FAIL) fail("GPU %d: %s", gpu, message.c_str());
    for (auto& v : verdicts)
      if (v.test == test && v.gpu == gpu) {
        if (health > v.health) v.health = health, v.value = value, v.message = message;
        return;
      }
    Verdict v{test, system, health, value, message};
    v.gpu = gpu;
    verdicts.push_back(v);
  }
  void verdict_fail(const std::string& test, uint32_t system, double value, const char* fmt, ...) __attribute__((format(printf, 5, 6))) {
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    fail("%s", buf);
    for (auto& v : verdicts)
      if (v.test == test && v.health == GM_HEALTH_FAIL) return;  // the first failure of a test is its verdict
    for (auto it = verdicts.begin(); it != verdicts.end();)
      it = it->test == test ? verdicts.erase(it) : it + 1;
    verdicts.push_back({test, system, GM_HEALTH_FAIL, value, buf});
  }
  void verdict_pass(const std::string& test, uint32_t system, double value, const std::string& message) {
    for (auto& v : verdicts)
      if (v.test == test) return;
    verdicts.push_back({test, system, GM_HEALTH_PASS, value, message});
  }
  // The whole report as one JSON object; `gpu` and `timestamp` let the exporter label and age it.
  std::string document(int gpu) const {
    std::string vj = "[";
    for (size_t i = 0; i < verdicts.size(); i++) {
      const Verdict& v = verdicts[i];
      char num[64];
      if (std::isfinite(v.value)) snprintf(num, sizeof num, "%.6g", v.value); else snprintf(num, sizeof num, "null");
      std::string msg;
      for (char c : v.message) {
        if (c == '"' || c == '\\') msg += '\\';
        msg += (unsigned char)c < 0x20 ? ' ' : c;
      }
      vj += std::string(i ? "," : "") + "{\"test\":\"" + v.test + "\"," + (v.gpu >= 0 ? "\"gpu\":" + std::to_string(v.gpu) + "," : "") +
            "\"system\":" + std::to_string(v.system) + ",\"health\":" +
            std::to_string(v.health) + ",\"value\":" + num + ",\"message\":\"" + msg + "\"}";
    }
    return "{\"gpu\":" + std::to_string(gpu) + ",\"timestamp\":" + std::to_string((long long)time(nullptr)) +
           (js.empty() ? "" : "," + js) + ",\"verdicts\":" + vj + "],\"result\":\"" + (failed ? "fail" : "pass") + "\"}";
  }
  int finish(int gpu, const std::string& publish_path) {
    const std::string doc = document(gpu);
    if (json) printf("%s\n", doc.c_str());
    else printf("\nResult: %s\n", failed ? "FAIL" : "PASS");
    if (!publish_path.empty()) {
      // temp + rename: the exporter never reads a half-written report
      const std::string tmp = publish_path + ".swp";
      FILE* f = fopen(tmp.c_str(), "w");
      if (!f || fputs(doc.c_str(), f) < 0 || fclose(f) != 0 || rename(tmp.c_str(), publish_path.c_str()) != 0) {
        fprintf(stderr, "error: cannot publish the report to %s: %s\n", publish_path.c_str(), strerror(errno));
        return 2;
      }
    }
    return failed ? 1 : 0;
  }
};

std::string strprintf_local(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
std::string strprintf_local(const char* fmt, ...) {
  char b[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof b, fmt, ap);
  va_end(ap);
  return b;
}

std::string jnum(double v) {
  if (std::isnan(v) || std::isinf(v)) return "null";
  char b[64];
  snprintf(b, sizeof b, "%.6g", v);
  return b;
}
std::string jstr(const std::string& s) {
  std::string o = "\"";
  for (char c : s) {
    if (c == '"' || c == '\\') o += '\\';
    o += c;
  }
  return o + "\"";
}

#define DIAG_CALL(rep, expr)                                                        \
  do {                                                                              \
    int _rc = (expr);                                                               \
    if (_rc != GMD_OK) {                                                            \
      fprintf(stderr, "error: %s -> %d: %s\n", #expr, _rc, gmd_last_error());       \
      return 2;                                                                     \
    }                                                                               \
  } while (0)

const char* kStreamNames[] = {"copy", "scale", "add", "triad", "read", "write"};
const char* kDtypeNames[] = {"bf16", "fp8", "f32", "mxfp8"};
struct PatName {
  const char* name;
  uint32_t bit;
};
const PatName kPatterns[] = {{"walking_ones", GMD_PAT_WALKING_ONES}, {"walking_zeros", GMD_PAT_WALKING_ZEROS},
                             {"address", GMD_PAT_ADDRESS},           {"checkerboard", GMD_PAT_CHECKERBOARD},
                             {"inv_checkerboard", GMD_PAT_INV_CHECKERBOARD}, {"random", GMD_PAT_RANDOM},
                             {"modulo_20", GMD_PAT_MODULO_20}};

int dtype_of(const std::string& s) {
  for (int i = 0; i < 4; i++)
    if (s == kDtypeNames[i]) return i;
  return -1;
}

// ------------------------------------------------------------------ commands --
int cmd_info(const Args& a, Report& r) {
  int n = gmd_device_count();
  if (n <= 0) {
    fprintf(stderr, "error: no HIP device: %s\n", gmd_last_error());
    return 2;
  }
  std::string arr = "[";
  for (int i = 0; i < n; i++) {
    gmd_device_info_t d;
    DIAG_CALL(r, gmd_device_info(i, &d));
    r.text("GPU %d: %s (%s) %s  CUs %d  %d MHz  HBM %.1f GiB (%.1f free)  L2 %d KiB  LDS/CU %d KiB\n", i, d.name, d.gcn_arch,
           d.pci_bdf, d.compute_units, d.clock_mhz, d.total_mem / 1073741824.0, d.free_mem / 1073741824.0,
           d.l2_bytes >> 10, d.lds_per_cu >> 10);
    arr += std::string(i ? "," : "") + "{\"gpu\":" + std::to_string(i) + ",\"name\":" + jstr(d.name) + ",\"arch\":" +
           jstr(d.gcn_arch) + ",\"bdf\":" + jstr(d.pci_bdf) + ",\"cus\":" + std::to_string(d.compute_units) +
           ",\"total_mem\":" + std::to_string(d.total_mem) + ",\"free_mem\":" + std::to_string(d.free_mem) + "}";
  }
  r.member("devices", arr + "]");
  return 0;
}

int cmd_hbm(const Args& a, Report& r) {
  const int gpu = (int)a.num("-g", 0);
  const uint64_t bytes = a.size("--bytes", 4ull << 30);
  const int iters = (int)a.num("--iters", 10);
  const std::string which = a.str("--kernel", "all");
  const double floor_tbps = a.num("--min-tbps", 0);
  r.text("HBM stream on GPU %d, %.1f GiB per buffer, %d iterations\n  %-8s %10s %10s %8s\n", gpu, bytes / 1073741824.0, iters,
         "kernel", "best TB/s", "median", "check");
  std::string obj = "{";
  bool any = false;
  double worst = 1e30;
  for (int k = 0; k < GMD_STREAM_KERNELS; k++) {
    if (which != "all" && which != kStreamNames[k]) continue;
    gmd_stream_result_t s;
    DIAG_CALL(r, gmd_hbm_stream(gpu, k, bytes, iters, 2, &s));
    r.text("  %-8s %10.3f %10.3f %8s\n", kStreamNames[k], s.tbps_best, s.tbps_median, s.verified ? "ok" : "WRONG");
    obj += std::string(any ? "," : "") + "\"" + kStreamNames[k] + "\":{\"tbps_best\":" + jnum(s.tbps_best) +
           ",\"tbps_median\":" + jnum(s.tbps_median) + ",\"verified\":" + (s.verified ? "true" : "false") + "}";
    any = true;
    if (!s.verified) r.verdict_fail("hbm_stream", GM_HEALTH_WATCH_MEM, s.tbps_best, "stream %s produced wrong data", kStreamNames[k]);
    if (floor_tbps > 0 && s.tbps_best < floor_tbps)
      r.verdict_fail("hbm_stream", GM_HEALTH_WATCH_MEM, s.tbps_best, "stream %s %.3f TB/s is under the %.3f TB/s floor", kStreamNames[k], s.tbps_best, floor_tbps);
    worst = std::min(worst, s.tbps_best);
  }
  if (any) r.verdict_pass("hbm_stream", GM_HEALTH_WATCH_MEM, worst, "every stream kernel verified");
  if (!any) {
    fprintf(stderr, "unknown kernel %s\n", which.c_str());
    return 2;
  }
  r.member("hbm_stream", obj + "}");
  return 0;
}

void progress_cb(double f, const char* what, void* user) {
  if (*(bool*)user) return;
  fprintf(stderr, "\r  %5.1f %%  %-24s", f * 100, what);
  if (f >= 1.0) fprintf(stderr, "\n");
}

int cmd_memtest(const Args& a, Report& r) {
  const int gpu = (int)a.num("-g", 0);
  gmd_memtest_config_t c;
  memset(&c, 0, sizeof c);
  c.bytes = a.size("--bytes", 0);
  c.fraction = a.num("--fraction", 0.90);
  c.chunk_bytes = a.size("--chunk", 0);
  c.passes = (uint32_t)a.num("--passes", 1);
  if (a.has("--patterns"))
    for (auto& p : csv(a.str("--patterns"))) {
      bool ok = false;
      for (auto& pn : kPatterns)
        if (p == pn.name) c.patterns |= pn.bit, ok = true;
      if (!ok) {
        fprintf(stderr, "unknown pattern %s\n", p.c_str());
        return 2;
      }
    }
  bool quiet = r.json || !isatty(2);
  c.progress = progress_cb;
  c.progress_user = &quiet;
  gmd_memtest_result_t m;
  DIAG_CALL(r, gmd_memtest(gpu, &c, &m));
  r.text("memtest on GPU %d: %.1f GiB in %u chunks, %u patterns, %.2f s\n  write %.2f TB/s  verify %.2f TB/s  errors %llu\n", gpu,
         m.bytes_tested / 1073741824.0, m.chunks, m.patterns_run, m.seconds, m.write_tbps, m.verify_tbps,
         (unsigned long long)m.errors);
  std::string log = "[";
  for (uint32_t i = 0; i < m.logged; i++) {
    r.text("  offset 0x%llx pattern 0x%x expected %016llx got %016llx\n", (unsigned long long)m.log[i].byte_offset,
           m.log[i].pattern, (unsigned long long)m.log[i].expected, (unsigned long long)m.log[i].actual);
    log += std::string(i ? "," : "") + "{\"offset\":" + std::to_string(m.log[i].byte_offset) + ",\"pattern\":" +
           std::to_string(m.log[i].pattern) + "}";
  }
  r.member("memtest", "{\"bytes_tested\":" + std::to_string(m.bytes_tested) + ",\"errors\":" + std::to_string(m.errors) +
                          ",\"patterns\":" + std::to_string(m.patterns_run) + ",\"write_tbps\":" + jnum(m.write_tbps) +
                          ",\"verify_tbps\":" + jnum(m.verify_tbps) + ",\"seconds\":" + jnum(m.seconds) + ",\"log\":" + log + "]}");
  if (m.errors) r.verdict_fail("memtest", GM_HEALTH_WATCH_MEM, (double)m.errors, "memtest found %llu bad words in %.1f GiB", (unsigned long long)m.errors, m.bytes_tested / 1073741824.0);
  else r.verdict_pass("memtest", GM_HEALTH_WATCH_MEM, 0, "no bad words");
  return 0;
}

int cmd_mfma(const Args& a, Report& r) {
  const int gpu = (int)a.num("-g", 0);
  const std::string which = a.str("--dtype", "all");
  const double secs = a.num("--seconds", 1.0);
  r.text("MFMA pipe peak on GPU %d (%.1f s per dtype)\n  %-6s %12s %12s %10s\n", gpu, secs, "dtype", "TFLOP/s", "best", "MHz");
  std::string obj = "{";
  bool any = false;
  for (int dt : {GMD_DT_BF16, GMD_DT_FP8, GMD_DT_MXFP8, GMD_DT_F32}) {
    if (which != "all" && which != kDtypeNames[dt]) continue;
    gmd_mfma_result_t m;
    DIAG_CALL(r, gmd_mfma_peak(gpu, dt, secs, &m));
    r.text("  %-6s %12.1f %12.1f %10.0f\n", kDtypeNames[dt], m.tflops, m.tflops_best, m.clock_mhz_est);
    obj += std::string(any ? "," : "") + "\"" + kDtypeNames[dt] + "\":{\"tflops\":" + jnum(m.tflops) + ",\"tflops_best\":" +
           jnum(m.tflops_best) + ",\"clock_mhz\":" + jnum(m.clock_mhz_est) + "}";
    any = true;
    if (m.verified < 0) r.verdict_fail("mfma_peak", GM_HEALTH_WATCH_COMPUTE, m.tflops, "MFMA %s pipe produced wrong results", kDtypeNames[dt]);
  }
  if (!any) return 2;
  r.verdict_pass("mfma_peak", GM_HEALTH_WATCH_COMPUTE, NAN, "matrix pipes ran");
  r.member("mfma_peak", obj + "}");
  return 0;
}

// Matrix-pipe activity during a soak, from the MFMA busy-cycle hardware counter.  The counter plugin cannot
// live in a process that runs HIP work (core/prof/prof_counters.h), so `gpumon-prof-probe` runs as a child and
// its lines are read from a pipe: "agent N: name  pci dddd:bb:dd.f ..." once, then per 250 ms window
// "agent N window Ws GRBM_NAME X (K instances) mfma_active Y clock_mhz Z ...".
struct MfmaWatch {
  pid_t pid = -1;
  int fd = -1;
  std::thread reader;
  std::map<int, std::string> bdf_of;             // agent -> pci address
  std::map<int, std::vector<double>> mfma, mhz;  // agent -> windows
  std::map<int, std::vector<double>> at;         // agent -> arrival time of each window (steady clock, s)
  std::mutex mu;
  static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
  // Average MFMA activity over the windows that lie wholly inside [t0, t1] (a window arrives when it ends and
  // is 250 ms long); -1 when there is none.
  double average_between(const char* pci_bdf, double t0, double t1, int* windows) {
    std::lock_guard<std::mutex> l(mu);
    double sum = 0;
    int n = 0;
    for (auto& kv : bdf_of) {
      if (strcasecmp(kv.second.c_str(), pci_bdf) != 0) continue;
      const auto &v = mfma[kv.first], &t = at[kv.first];
      for (size_t i = 0; i < v.size() && i < t.size(); i++)
        if (t[i] - 0.26 >= t0 && t[i] <= t1 + 0.02) sum += v[i], n++;
    }
    *windows = n;
    return n ? sum / n : -1.0;
  }
  bool start() {
    char self[4096];
    ssize_t n = readlink("/proc/self/exe", self, sizeof self - 1);
    if (n <= 0) return false;
    std::string probe(self, (size_t)n);
    probe = probe.substr(0, probe.rfind('/')) + "/gpumon-prof-probe";
    if (access(probe.c_str(), X_OK) != 0) return false;
    int pfd[2];
    if (pipe(pfd) != 0) return false;
    posix_spawn_file_actions_t fa;
    posix_spawn_file_actions_init(&fa);
    posix_spawn_file_actions_adddup2(&fa, pfd[1], 1);
    posix_spawn_file_actions_addclose(&fa, pfd[0]);
    posix_spawn_file_actions_addclose(&fa, pfd[1]);
    const char* argv[] = {probe.c_str(), "--interval-ms", "250", "--count", "1000000", nullptr};
    const int rc = posix_spawn(&pid, probe.c_str(), &fa, nullptr, const_cast<char* const*>(argv), environ);
    posix_spawn_file_actions_destroy(&fa);
    close(pfd[1]);
    if (rc != 0) {
      close(pfd[0]);
      pid = -1;
      return false;
    }
    fd = pfd[0];
    reader = std::thread([this] {
      FILE* f = fdopen(fd, "r");
      if (!f) return;
      char line[1024];
      while (fgets(line, sizeof line, f)) {
        int agent = 0;
        char name[64], bdf[32], grbm[64];
        int inst = 0;
        double w, ga, ma, mz;
        std::lock_guard<std::mutex> l(mu);
        if (sscanf(line, "agent %d: %63s pci %31s", &agent, name, bdf) == 3) bdf_of[agent] = bdf;
        else if (sscanf(line, "agent %d window %lfs %63s %lf (%d instances) mfma_active %lf clock_mhz %lf", &agent, &w, grbm, &ga, &inst, &ma, &mz) == 7) {
          mfma[agent].push_back(ma);
          mhz[agent].push_back(mz);
          at[agent].push_back(now());
        }
      }
      fclose(f);
    });
    return true;
  }
  // Ends the child and folds the windows of the device with this PCI address; the first window after the
  // start and the last one (cut short by the end of the soak) are left out when there are enough.
  std::string finish(Report& r, const char* pci_bdf) {
    if (pid < 0) return "null";
    kill(pid, SIGTERM);
    int st = 0;
    waitpid(pid, &st, 0);
    pid = -1;
    if (reader.joinable()) reader.join();
    for (auto& kv : bdf_of) {
      if (strcasecmp(kv.second.c_str(), pci_bdf) != 0) continue;
      std::vector<double> v = mfma[kv.first], c = mhz[kv.first];
      if (v.size() > 4) v = std::vector<double>(v.begin() + 1, v.end() - 1), c = std::vector<double>(c.begin() + 1, c.end() - 1);
      if (v.empty()) break;
      auto fold = [](const std::vector<double>& x, double* lo, double* avg, double* hi) {
        *lo = 1e300, *hi = -1e300, *avg = 0;
        for (double e : x) *lo = std::min(*lo, e), *hi = std::max(*hi, e), *avg += e / x.size();
      };
      double lo, avg, hi, clo, cavg, chi;
      fold(v, &lo, &avg, &hi);
      fold(c, &clo, &cavg, &chi);
      r.text("  matrix pipes (MFMA busy counter, 250 ms windows): min %5.1f  avg %5.1f  max %5.1f %% busy at %4.0f MHz  (%d windows)\n", lo,
             avg, hi, cavg, (int)v.size());
      return "{\"min\":" + jnum(lo) + ",\"avg\":" + jnum(avg) + ",\"max\":" + jnum(hi) + ",\"clock_mhz\":" + jnum(cavg) +
             ",\"windows\":" + std::to_string(v.size()) + "}";
    }
    r.text("  matrix pipes: the counter probe gave no windows for this device\n");
    return "null";
  }
};

// Engine-side telemetry of the GPU under stress: power / temperature / clocks at 100 ms.
struct Soak {
  bool on = false;
  uint32_t gpu = 0;
  int64_t t0 = 0;
  gm_group_t group = 0;
  gm_field_group_t fg = 0;
  bool start(int hip_dev) {
    gmd_device_info_t d;
    if (gmd_device_info(hip_dev, &d) != GMD_OK) return false;
    gm_init_params_t p;
    memset(&p, 0, sizeof p);
    p.version = GM_INIT_PARAMS_VERSION;
    p.mode = GM_MODE_EMBEDDED;
    if (gm_init(&p) != GM_ST_OK) return false;
    uint32_t n = 0;
    gm_device_count(&n);
    bool found = false;
    for (uint32_t i = 0; i < n && !found; i++) {
      gm_device_attributes_t at;
      memset(&at, 0, sizeof at);
      at.version = GM_DEVICE_ATTRIBUTES_VERSION;
      if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, d.pci_bdf) == 0) gpu = i, found = true;
    }
    if (!found) {
      gm_shutdown();
      return false;
    }
    uint16_t ids[] = {GM_FI_POWER_USAGE, GM_FI_GPU_TEMP, GM_FI_MEM_TEMP, GM_FI_GFX_CLOCK, GM_FI_MEM_CLOCK, GM_FI_GFX_UTIL};
    gm_group_create("diag-soak", &group);
    gm_group_add_device(group, gpu);
    gm_field_group_create(ids, 6, "diag-soak", &fg);
    gm_watch_fields(group, fg, 100000, 600.0, 0);
    gm_update_all_fields(1);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    t0 = (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
    on = true;
    return true;
  }
  std::string summarize(Report& r) {
    if (!on) return "null";
    struct F {
      uint16_t id;
      const char* name;
      const char* unit;
    } fs[] = {{GM_FI_POWER_USAGE, "power", "W"},     {GM_FI_GPU_TEMP, "gpu_temp", "C"},   {GM_FI_MEM_TEMP, "hbm_temp", "C"},
              {GM_FI_GFX_CLOCK, "gfx_clock", "MHz"}, {GM_FI_MEM_CLOCK, "mem_clock", "MHz"}, {GM_FI_GFX_UTIL, "gfx_util", "%"}};
    std::string obj = "{";
    r.text("  telemetry during the run (100 ms samples):\n");
    bool first = true;
    for (auto& f : fs) {
      std::vector<gm_value_t> v(8192);
      int n = (int)v.size();
      if (gm_values_since(gpu, f.id, t0, v.data(), &n) != GM_ST_OK) continue;
      double lo = 1e300, hi = -1e300, sum = 0;
      int cnt = 0;
      for (int i = 0; i < n; i++) {
        if (v[i].status != GM_VS_OK) continue;
        double x = v[i].type == GM_FT_DOUBLE ? v[i].v.dbl : (double)v[i].v.i64;
        lo = std::min(lo, x), hi = std::max(hi, x), sum += x, cnt++;
      }
      if (!cnt) continue;
      r.text("    %-10s min %8.1f  avg %8.1f  max %8.1f %s  (%d samples)\n", f.name, lo, sum / cnt, hi, f.unit, cnt);
      obj += std::string(first ? "" : ",") + "\"" + f.name + "\":{\"min\":" + jnum(lo) + ",\"avg\":" + jnum(sum / cnt) +
             ",\"max\":" + jnum(hi) + ",\"samples\":" + std::to_string(cnt) + "}";
      first = false;
    }
    return obj + "}";
  }
  void stop() {
    if (on) gm_shutdown();
    on = false;
  }
};

int cmd_gemm(const Args& a, Report& r) {
  const int gpu = (int)a.num("-g", 0);
  const std::string which = a.str("--dtype", "bf16,mxfp8");
  const int m = (int)a.num("--m", 8192), n = (int)a.num("--n", 8192), k = (int)a.num("--k", 8192);
  const double secs = a.num("--seconds", 5.0), floor_tf = a.num("--min-tflops", 0);
  MfmaWatch mfma;  // started first: a child process, before this one has touched the device
  const bool want_mfma = a.has("--mfma-counters") && mfma.start();
  if (a.has("--mfma-counters") && !want_mfma) fprintf(stderr, "note: gpumon-prof-probe not found next to this binary; no matrix-pipe counters\n");
  Soak soak;
  if (!a.has("--no-telemetry") && !soak.start(gpu)) fprintf(stderr, "note: engine telemetry unavailable; running without it\n");
  std::string obj = "{";
  bool any = false;
  auto end_watch = [&] {
    gmd_device_info_t d;
    if (want_mfma) r.member("mfma_active", mfma.finish(r, gmd_device_info(gpu, &d) == GMD_OK ? d.pci_bdf : ""));
  };
  for (auto& name : csv(which)) {
    int dt = dtype_of(name);
    if (dt < 0 || dt == GMD_DT_F32) {
      fprintf(stderr, "unknown GEMM dtype %s\n", name.c_str());
      soak.stop();
      end_watch();
      return 2;
    }
    gmd_mfma_result_t g;
    const double t_begin = MfmaWatch::now();
    int rc = gmd_gemm_stress(gpu, dt, m, n, k, secs, 1, &g);
    const double t_end = MfmaWatch::now();
    if (rc != GMD_OK && rc != GMD_ERR_VERIFY) {
      fprintf(stderr, "error: gemm stress -> %d: %s\n", rc, gmd_last_error());
      soak.stop();
      end_watch();
      return 2;
    }
    r.text("GEMM stress %s %dx%dx%d on GPU %d: %.1f TFLOP/s (best %.1f), %llu launches in %.1f s, sdc mismatches %llu, "
           "max rel err vs fp32 reference %.2e, check %s\n",
           name.c_str(), m, n, k, gpu, g.tflops, g.tflops_best, (unsigned long long)g.launches, g.seconds,
           (unsigned long long)g.sdc_mismatches, g.max_rel_err, g.verified > 0 ? "ok" : g.verified < 0 ? "WRONG" : "skipped");
    obj += std::string(any ? "," : "") + "\"" + name + "\":{\"tflops\":" + jnum(g.tflops) + ",\"tflops_best\":" +
           jnum(g.tflops_best) + ",\"launches\":" + std::to_string(g.launches) + ",\"sdc_mismatches\":" +
           std::to_string(g.sdc_mismatches) + ",\"max_rel_err\":" + jnum(g.max_rel_err) + ",\"verified\":" +
           std::to_string(g.verified);
    if (want_mfma) {  // the windows that lie inside this dtype's soak
      usleep(300000);  // let the window that ends with the soak arrive
      gmd_device_info_t d;
      int wn = 0;
      const double avg = gmd_device_info(gpu, &d) == GMD_OK ? mfma.average_between(d.pci_bdf, t_begin, t_end, &wn) : -1.0;
      if (wn > 0) {
        r.text("  matrix pipes during the %s soak: %.1f %% busy (MFMA busy counter, %d windows)\n", name.c_str(), avg, wn);
        obj += ",\"mfma_active_avg\":" + jnum(avg);
      }
    }
    obj += "}";
    any = true;
    const std::string test = "gemm_" + name;
    if (g.verified < 0 && !g.sdc_mismatches)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, g.max_rel_err,
                     "GEMM %s: sampled entries differ from the fp32 reference (max relative error %.3g, tolerance 4e-3)", name.c_str(), g.max_rel_err);
    if (g.sdc_mismatches)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, (double)g.sdc_mismatches,
                     "GEMM %s: %llu launches changed their checksum (silent data corruption)", name.c_str(), (unsigned long long)g.sdc_mismatches);
    if (floor_tf > 0 && g.tflops < floor_tf)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, g.tflops, "GEMM %s %.1f TFLOP/s is under the %.1f floor", name.c_str(), g.tflops, floor_tf);
    r.verdict_pass(test, GM_HEALTH_WATCH_COMPUTE, g.tflops, strprintf_local("%.1f TFLOP/s, reference and checksums agree", g.tflops));
  }
  r.member("gemm_stress", obj + "}");
  r.member("telemetry", soak.summarize(r));
  soak.stop();
  end_watch();
  return 0;
}

bool parse_gpus(const Args& a, std::vector<int>* ids) {
  if (!a.has("--gpus")) return true;
  for (auto& s : csv(a.str("--gpus"))) ids->push_back(atoi(s.c_str()));
  return !ids->empty();
}

int run_sweep(const Args& a, Report& r, const std::vector<int>& ids, std::string* js) {
  gmd_rccl_config_t c;
  memset(&c, 0, sizeof c);
  c.ndev = (int)ids.size();
  c.devices = ids.empty() ? nullptr : ids.data();
  c.min_bytes = a.size("--min-bytes", 8);
  c.max_bytes = a.size("--max-bytes", 1ull << 30);
  c.factor = (int)a.num("--factor", 4);
  c.iters = (int)a.num("--iters", 0);
  c.verify = a.has("--no-verify") ? 0 : 1;
  if (a.has("--ops"))
    for (auto& o : csv(a.str("--ops"))) {
      bool ok = false;
      for (int i = 0; i < GMD_COLL_COUNT; i++)
        if (o == gmd_collective_name(i)) c.ops_mask |= 1u << i, ok = true;
      if (!ok) {
        fprintf(stderr, "unknown collective %s\n", o.c_str());
        return 2;
      }
    }
  std::vector<gmd_rccl_point_t> pts(512);
  int n = (int)pts.size();
  DIAG_CALL(r, gmd_rccl_sweep(&c, pts.data(), &n));
  double peak[GMD_COLL_COUNT] = {0};
  int w = n ? pts[0].ndev : 0;
  r.text("# RCCL sweep on %d GPU%s\n# %-15s%14s%7s%12s%13s%13s  check\n", w, w == 1 ? "" : "s", "op", "bytes", "iters", "time(us)",
         "algbw(GB/s)", "busbw(GB/s)");
  // gmd_rccl_sweep fails unless every communicator reports exactly the requested rank count on its own device
  *js += "{\"gpus\":" + std::to_string(w) + ",\"communicator_ranks\":" + std::to_string(w) + ",\"points\":[";
  for (int i = 0; i < n; i++) {
    auto& p = pts[i];
    r.text("  %-15s%14llu%7d%12.1f%13.2f%13.2f  %s\n", gmd_collective_name(p.op), (unsigned long long)p.bytes, p.iters, p.time_us,
           p.algbw_gbps, p.busbw_gbps, p.verified > 0 ? "ok" : p.verified < 0 ? "MISMATCH" : "-");
    peak[p.op] = std::max(peak[p.op], p.busbw_gbps);
    *js += std::string(i ? "," : "") + "{\"op\":" + jstr(gmd_collective_name(p.op)) + ",\"bytes\":" + std::to_string(p.bytes) +
           ",\"time_us\":" + jnum(p.time_us) + ",\"algbw_gbps\":" + jnum(p.algbw_gbps) + ",\"busbw_gbps\":" + jnum(p.busbw_gbps) +
           ",\"verified\":" + std::to_string(p.verified) + "}";
    if (p.verified < 0)
      r.verdict_fail("rccl", GM_HEALTH_WATCH_XGMI, (double)p.bytes, "%s of %llu bytes on %d GPUs returned wrong data", gmd_collective_name(p.op),
                     (unsigned long long)p.bytes, w);
  }
  *js += "],\"peak_busbw_gbps\":{";
  r.text("# peak busbw (GB/s):");
  bool first = true;
  for (int i = 0; i < GMD_COLL_COUNT; i++) {
    if (peak[i] <= 0) continue;
    r.text("  %s=%.1f", gmd_collective_name(i), peak[i]);
    *js += std::string(first ? "" : ",") + "\"" + gmd_collective_name(i) + "\":" + jnum(peak[i]);
    first = false;
  }
  r.text("\n");
  *js += "}}";
  // A floor only means something once links are involved: the 1-GPU row is a local copy.
  const double floor_gbps = a.num("--min-busbw", 0);
  if (floor_gbps > 0 && w > 1)
    for (int i = 0; i < GMD_COLL_COUNT; i++)
      if (peak[i] > 0 && peak[i] < floor_gbps)
        r.verdict_fail("rccl", GM_HEALTH_WATCH_XGMI, peak[i], "%s peaks at %.1f GB/s bus bandwidth on %d GPUs, under the %.1f GB/s floor",
                       gmd_collective_name(i), peak[i], w, floor_gbps);
  r.verdict_pass("rccl", GM_HEALTH_WATCH_XGMI, NAN, "every collective returned the right data");
  return 0;
}

int cmd_rccl(const Args& a, Report& r) {
  std::vector<int> ids;
  if (!parse_gpus(a, &ids)) return 2;
  if (!a.has("--scaling")) {
    std::string js;
    int rc = run_sweep(a, r, ids, &js);
    if (rc) return rc;
    r.member("rccl", js);
    return 0;
  }
  // 1 / 2 / 4 / 8 ... GPUs: the scaling series.
  int visible = gmd_device_count();
  if (ids.empty())
    for (int i = 0; i < visible; i++) ids.push_back(i);
  std::string arr = "[";
  for (size_t w = 1; w <= ids.size(); w *= 2) {
    std::vector<int> sub(ids.begin(), ids.begin() + w);
    std::string js;
    int rc = run_sweep(a, r, sub, &js);
    if (rc) return rc;
    arr += std::string(w > 1 ? "," : "") + js;
  }
  r.member("rccl_scaling", arr + "]");
  return 0;
}

// Which physical link of a GPU leads to which peer, from the engine's link table (matched by PCI address, since
// the compute runtime and the engine may number the cards differently).
struct LinkMap {
  bool on = false;
  std::vector<int> engine_of;  // device-list position -> engine GPU index, -1 unknown
  std::vector<std::vector<gm_xgmi_link_t>> links;  // per position
  double per_direction_gbytes = 0;  // advertised rate of one direction of one link, GB/s; 0 unknown

  void load(const std::vector<int>& hip_ids) {
    gm_init_params_t p;
    memset(&p, 0, sizeof p);
    p.version = GM_INIT_PARAMS_VERSION;
    p.mode = GM_MODE_EMBEDDED;
    if (gm_init(&p) != GM_ST_OK) return;
    on = true;
    uint32_t n = 0;
    gm_device_count(&n);
    for (int hip : hip_ids) {
      int idx = -1;
      gmd_device_info_t d;
      if (gmd_device_info(hip, &d) == GMD_OK)
        for (uint32_t i = 0; i < n && idx < 0; i++) {
          gm_device_attributes_t at;
          memset(&at, 0, sizeof at);
          at.version = GM_DEVICE_ATTRIBUTES_VERSION;
          if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, d.pci_bdf) == 0) idx = (int)i;
        }
      engine_of.push_back(idx);
      std::vector<gm_xgmi_link_t> l(64);
      int cnt = (int)l.size();
      if (idx < 0 || gm_xgmi_links((uint32_t)idx, l.data(), &cnt) != GM_ST_OK) cnt = 0;
      l.resize(cnt);
      for (auto& x : l)
        if (x.max_bandwidth_gbps > 0) per_direction_gbytes = std::max(per_direction_gbytes, x.max_bandwidth_gbps / 8.0);
      links.push_back(l);
    }
    gm_shutdown();
  }
  // Link index on position i whose far end is position j; -1 when the platform does not say.
  int link_to(int i, int j) const {
    if (!on || i >= (int)links.size() || j >= (int)engine_of.size() || engine_of[j] < 0) return -1;
    for (auto& x : links[i])
      if (x.peer_gpu == engine_of[j]) return (int)x.link;
    return -1;
  }
};

// xGMI link test: the peer-store matrix in three traffic patterns, graded per link.
int cmd_links(const Args& a, Report& r) {
  std::vector<int> ids;
  if (!parse_gpus(a, &ids)) return 2;
  if (ids.empty())
    for (int i = 0; i < gmd_device_count(); i++) ids.push_back(i);
  const int w = (int)ids.size();
  if (w <= 0) {
    fprintf(stderr, "error: no GPU visible\n");
    return 2;
  }
  if (w == 1) {
    r.text("# xGMI link test: one GPU visible, no link to test\n");
    r.member("links", "{\"gpus\":1,\"tested\":0}");
    r.member("p2p_gbps", "[[0]]");
    return 0;
  }
  LinkMap map;
  map.load(ids);
  // Absolute floor for one direction of one link: half the advertised rate unless given.  (The advertised rate is
  // lanes x bit rate; protocol overhead and the store path keep a healthy link well above half of it.)
  double floor = a.has("--floor-gbps") ? a.num("--floor-gbps", 0) : 0.5 * (map.per_direction_gbytes > 0 ? map.per_direction_gbytes : 76.0);
  gmd_link_rule_t rule = {floor, a.num("--warn-ratio", 0), a.num("--fail-ratio", 0)};

  gmd_link_config_t c;
  memset(&c, 0, sizeof c);
  c.ndev = w;
  c.devices = ids.data();
  c.bytes = a.size("--bytes", 256ull << 20);
  c.iters = (int)a.num("--iters", 5);
  c.verify = a.has("--no-verify") ? 0 : 1;
  c.workgroups = (int)a.num("--workgroups", 0);

  const char* names[] = {"uni", "bidi", "all"};
  const char* what[] = {"one pair at a time", "both directions of a pair at once", "every GPU to every peer at once"};
  std::vector<std::string> modes = a.has("--modes") ? csv(a.str("--modes")) : std::vector<std::string>{"uni", "bidi", "all"};
  std::string js = "{\"gpus\":" + std::to_string(w) + ",\"floor_gbps\":" + jnum(floor);
  std::vector<int> worst(w, GM_HEALTH_PASS);
  std::vector<std::string> worst_msg(w);
  std::vector<double> slowest(w, 1e300);
  for (int mode = 0; mode < 3; mode++) {
    if (std::find(modes.begin(), modes.end(), names[mode]) == modes.end()) continue;
    std::vector<double> m((size_t)w * w);
    std::vector<uint64_t> bad((size_t)w * w);
    DIAG_CALL(r, gmd_link_test(&c, mode, m.data(), bad.data()));
    r.text("# link bandwidth, GB/s, %s (row = source GPU)\n", what[mode]);
    std::string mj = "[";
    for (int i = 0; i < w; i++) {
      mj += std::string(i ? "," : "") + "[";
      r.text("  ");
      for (int j = 0; j < w; j++) {
        r.text("%8.1f", m[(size_t)i * w + j]);
        mj += std::string(j ? "," : "") + jnum(m[(size_t)i * w + j]);
      }
      r.text("\n");
      mj += "]";
    }
    js += std::string(",\"") + names[mode] + "\":" + mj + "]";
    if (mode == GMD_LINK_UNI) r.member("p2p_gbps", mj + "]");
    // The floor is one direction's; with both directions or all links busy a healthy link keeps most of it.
    gmd_link_rule_t mr = rule;
    mr.floor_gbps = mode == GMD_LINK_UNI ? floor : mode == GMD_LINK_BIDI ? 0.8 * floor : 0;
    std::vector<gmd_link_verdict_t> v((size_t)w * (w - 1));
    int nv = (int)v.size();
    int rc = gmd_link_grade(w, m.data(), c.verify ? bad.data() : nullptr, &mr, v.data(), &nv);
    if (rc < 0) return 2;
    for (int k = 0; k < nv; k++) {
      const int link = map.link_to(v[k].src, v[k].dst);
      std::string where = "link " + (link >= 0 ? std::to_string(link) : std::string("?")) + " to GPU " + std::to_string(ids[v[k].dst]) +
                          " (" + names[mode] + "): " + v[k].message;
      if (v[k].health != GM_HEALTH_PASS) r.text("  %s GPU %d %s\n", v[k].health == GM_HEALTH_FAIL ? "FAIL" : "WARN", ids[v[k].src], where.c_str());
      if (v[k].gbps >= 0) slowest[v[k].src] = std::min(slowest[v[k].src], v[k].gbps);
      if (v[k].health > worst[v[k].src]) worst[v[k].src] = v[k].health, worst_msg[v[k].src] = where;
    }
  }
  // link -> peer table, so that a verdict can be traced to a cable
  std::string lj = "[";
  bool any = false;
  for (int i = 0; i < w; i++)
    for (int j = 0; j < w; j++) {
      const int link = i == j ? -1 : map.link_to(i, j);
      if (link < 0) continue;
      lj += std::string(any ? "," : "") + "{\"gpu\":" + std::to_string(ids[i]) + ",\"link\":" + std::to_string(link) + ",\"peer\":" + std::to_string(ids[j]) + "}";
      any = true;
    }
  js += ",\"link_peers\":" + lj + "]}";
  r.member("links", js);
  for (int i = 0; i < w; i++)
    r.verdict_for_gpu(ids[i], "xgmi_p2p", GM_HEALTH_WATCH_XGMI, worst[i], slowest[i] < 1e300 ? slowest[i] : std::nan(""),
                      worst[i] == GM_HEALTH_PASS ? strprintf_local("%d links, slowest %.1f GB/s", w - 1, slowest[i]) : worst_msg[i]);
  return 0;
}

int cmd_run(const Args& a, Report& r) {
  const int level = (int)a.num("-r", 1);
  if (level < 1 || level > 3) {
    fprintf(stderr, "-r takes 1, 2 or 3\n");
    return 2;
  }
  Args sub = a;
  int rc;
  if ((rc = cmd_info(sub, r))) return rc;
  // level 1 (quick, seconds): short stream + 1 GiB memtest.  level 2 adds the matrix pipes and a 5 s
  // GEMM stress.  level 3 sweeps 90 % of HBM, soaks 60 s and exercises the fabric.
  sub.kv["--bytes"] = level == 1 ? "1G" : "4G";
  sub.kv["--iters"] = level == 1 ? "5" : "20";
  if ((rc = cmd_hbm(sub, r))) return rc;
  if (level < 3) sub.kv["--bytes"] = level == 1 ? "1G" : "16G";
  else sub.kv.erase("--bytes");
  if ((rc = cmd_memtest(sub, r))) return rc;
  if (level >= 2) {
    sub.kv["--seconds"] = "1";
    if ((rc = cmd_mfma(sub, r))) return rc;
    sub.kv["--seconds"] = level == 2 ? "5" : "60";
    sub.kv.erase("--dtype");
    if ((rc = cmd_gemm(sub, r))) return rc;
  }
  if (level >= 3) {
    sub.kv.erase("--bytes");
    sub.kv.erase("--iters");
    sub.kv["--scaling"] = "1";
    if ((rc = cmd_rccl(sub, r))) return rc;
    if (gmd_device_count() > 1 && (rc = cmd_links(sub, r))) return rc;
  }
  return 0;
}

void usage() {
  fprintf(stderr,
          "usage: gpumon-diag <info|hbm|memtest|mfma|gemm|rccl|links|run> [options] [--json] [--publish FILE]\n"
          "  --publish FILE leaves the JSON report where `gpumon-exporter --diag-results` picks it up\n"
          "  hbm     [-g GPU] [--bytes N] [--iters N] [--kernel copy|scale|add|triad|read|write|all] [--min-tbps X]\n"
          "  memtest [-g GPU] [--bytes N | --fraction F] [--chunk N] [--patterns LIST] [--passes N]\n"
          "  mfma    [-g GPU] [--dtype bf16|fp8|mxfp8|f32|all] [--seconds S]\n"
          "  gemm    [-g GPU] [--dtype LIST] [--m M --n N --k K] [--seconds S] [--min-tflops X] [--no-telemetry] [--mfma-counters]\n"
          "  rccl    [--gpus LIST] [--ops LIST] [--min-bytes N] [--max-bytes N] [--factor N] [--iters N] [--scaling] [--no-verify] [--min-busbw GBPS]\n"
          "  links   [--gpus LIST] [--bytes N] [--iters N] [--modes uni,bidi,all] [--floor-gbps X] [--warn-ratio R] [--fail-ratio R]\n"
          "          peer-store bandwidth of every xGMI link, graded against a floor and against the other links\n"
          "  run     -r 1|2|3 [-g GPU]\n"
          "every command: [--json] [--publish FILE] [--submit ENGINE_ADDRESS [--submit-unix]]\n");
}

// --submit ADDRESS [--submit-unix]: hand the verdicts to a running gpumon-hostengine, where a failed test becomes a
// health incident of its subsystem and a GPU event (gm_diag_result_submit).  The engine numbers GPUs in PCI order,
// the compute runtime may not: the GPU is looked up by its PCI address.
bool submit_verdicts(const Args& a, const Report& r) {
  const std::string addr = a.str("--submit");
  gm_init_params_t p;
  memset(&p, 0, sizeof p);
  p.version = GM_INIT_PARAMS_VERSION;
  p.mode = GM_MODE_STANDALONE;
  p.address = addr.c_str();
  p.address_is_unix_socket = a.has("--submit-unix") || addr.find('/') != std::string::npos;
  gm_status_t st = gm_init(&p);
  if (st != GM_ST_OK) {
    fprintf(stderr, "error: cannot reach the host engine at %s: %s\n", addr.c_str(), gm_strerror(st));
    return false;
  }
  // HIP device index -> the engine's index of the same card
  auto engine_index = [](uint32_t hip) {
    uint32_t n = 0;
    gmd_device_info_t d;
    if (gmd_device_info((int)hip, &d) == GMD_OK && gm_device_count(&n) == GM_ST_OK)
      for (uint32_t i = 0; i < n; i++) {
        gm_device_attributes_t at;
        memset(&at, 0, sizeof at);
        at.version = GM_DEVICE_ATTRIBUTES_VERSION;
        if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, d.pci_bdf) == 0) return i;
      }
    return hip;
  };
  const uint32_t gpu = engine_index((uint32_t)a.num("-g", 0));
  bool ok = true;
  for (auto& v : r.verdicts) {
    gm_diag_result_t dr;
    memset(&dr, 0, sizeof dr);
    dr.version = GM_DIAG_RESULT_VERSION;
    dr.gpu = v.gpu >= 0 ? engine_index((uint32_t)v.gpu) : gpu;
    snprintf(dr.test, sizeof dr.test, "%s", v.test.c_str());
    dr.system = v.system;
    dr.health = v.health;
    dr.value = v.value;
    snprintf(dr.message, sizeof dr.message, "%s", v.message.c_str());
    st = gm_diag_result_submit(&dr);
    if (st != GM_ST_OK) {
      fprintf(stderr, "error: the host engine refused the %s verdict: %s\n", v.test.c_str(), gm_strerror(st));
      ok = false;
    }
  }
  gm_shutdown();
  return ok;
}

}  // namespace

int main(int argc, char** argv) {
  if (argc < 2) {
    usage();
    return 2;
  }
  std::string cmd = argv[1];
  Args a;
  if (!parse(std::vector<std::string>(argv + 2, argv + argc), &a)) {
    usage();
    return 2;
  }
  Report r;
  r.json = a.has("--json");
  int rc;
  if (cmd == "info") rc = cmd_info(a, r);
  else if (cmd == "hbm") rc = cmd_hbm(a, r);
  else if (cmd == "memtest") rc = cmd_memtest(a, r);
  else if (cmd == "mfma") rc = cmd_mfma(a, r);
  else if (cmd == "gemm") rc = cmd_gemm(a, r);
  else if (cmd == "rccl") rc = cmd_rccl(a, r);
  else if (cmd == "links" || cmd == "p2p") rc = cmd_links(a, r);
  else if (cmd == "run") rc = cmd_run(a, r);
  else {
    usage();
    return 2;
  }
  if (rc) return rc;
  rc = r.finish((int)a.num("-g", 0), a.str("--publish"));
  if (rc != 2 && a.has("--submit") && !submit_verdicts(a, r)) return 2;
  return rc;
}
