// gpumon-diag — active diagnostics for MI355X nodes (the `dcgmi diag` role; the
// reference itself ships no active test, only the passive health watches of
// bindings/go/dcgm/health.go:55-119).
//
//   gpumon-diag info                       HIP-side device facts
//   gpumon-diag hbm    [-g GPU | --gpus LIST|all] [--bytes N] [--iters N] [--kernel NAME|all] [--min-tbps X]
//   gpumon-diag memtest[-g GPU | --gpus LIST|all] [--bytes N | --fraction F] [--patterns a,b] [--passes N] [--march]
//   gpumon-diag mfma   [-g GPU | --gpus LIST|all] [--dtype bf16|fp8|mxfp8|f32|all] [--seconds S]
//   gpumon-diag gemm   [-g GPU | --gpus LIST|all] [--dtype bf16|fp8|mxfp8] [--m M --n N --k K] [--seconds S] [--check-every N]
//   gpumon-diag rccl   [--gpus 0,1,..] [--ops a,b] [--min-bytes N] [--max-bytes N] [--scaling] [--min-busbw X]
//   gpumon-diag links  [--gpus 0,1,..] [--bytes N] [--modes uni,bidi,all] [--direction push|pull|both]   (alias: p2p)
//   gpumon-diag run -r 1|2|3 [-g GPU | --gpus LIST|all] [--no-fabric]   graded suites: quick / medium / long
// Grading: every throughput is held against what a healthy GPU of this model delivers (cmd/diag_expect.h):
// PASS, WARN below 90 %, FAIL below 75 %; --expect key=value,.. replaces entries, --expect-scale F scales them,
// --no-expect grades correctness only.  With several GPUs every GPU runs the test at the same time (one host thread
// and stream set per GPU) and is also held against the median of the others.
// Every command takes --json and --publish FILE (leave the JSON report where
// `gpumon-exporter --diag-results` picks it up and exports it as gpumon_diag_* metrics).
// Exit status: 0 pass (warnings included), 1 a test failed, 2 usage / setup error.
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
#include <functional>
#include <map>
#include <mutex>
#include <atomic>
#include <thread>

#include "../diag/gpumon_diag.h"
#include "cli_common.h"
#include "diag_expect.h"

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
                                 "--warn-ratio", "--fail-ratio", "--workgroups", "--submit", "--min-busbw", "--expect", "--expect-scale",
                                 "--check-every", "--direction", "--link-gbps", "--outlier-warn", "--outlier-fail", "--arch", "--name", "--check",
                                 "--pcie-gen", "--pcie-width", "--backend", "--expect-gpus"};

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

// Collected output: a text report and a JSON document built side by side.  A report is about one GPU, or about
// the node (fabric tests, several GPUs at once): then it holds one sub-report per GPU.
struct Report {
  bool json = false;
  bool failed = false, warned = false;
  bool buffered = false;  // text goes to `out` instead of stdout (a GPU's thread of a concurrent run)
  std::string out;
  std::string js;       // comma-joined members of the top-level object
  int gpu = -1;         // HIP device this report is about; -1 = the node
  std::string pci_bdf;  // of that device: what the engine and the exporter find the card by
  std::map<std::string, double> metrics;  // measured rates by expectation key, for the comparison between GPUs
  std::vector<Report> subs;               // per-GPU reports of a node report

  void text(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    if (json) return;
    char b[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    if (buffered) {
      out += b;
      return;
    }
    fputs(b, stdout);
    fflush(stdout);
  }
  void member(const std::string& key, const std::string& value) {
    if (!js.empty()) js += ",";
    js += "\"" + key + "\":" + value;
  }
  void complain(int health, const std::string& msg) {
    if (health >= GM_HEALTH_FAIL) failed = true;
    else if (health >= GM_HEALTH_WARN) warned = true;
    else return;
    const std::string line = std::string(health >= GM_HEALTH_FAIL ? "FAIL: " : "WARN: ") + msg + "\n";
    if (buffered) out += json ? "" : "  " + line;
    else fputs(line.c_str(), stderr);
  }
  void fail(const char* fmt, ...) __attribute__((format(printf, 2, 3))) {
    char b[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(b, sizeof b, fmt, ap);
    va_end(ap);
    complain(GM_HEALTH_FAIL, b);
  }
  // Per-test verdicts, the part of the report the host engine understands (gm_diag_result_t).  One verdict per
  // (test, GPU): the worst health a test reached is its verdict, and a pass is recorded only where the test has
  // nothing worse, so that a later clean run replaces an old failure in the engine.
  struct Verdict {
    std::string test;
    uint32_t system;  // GM_HEALTH_WATCH_* bit
    int health;       // GM_HEALTH_PASS / WARN / FAIL
    double value;
    std::string message;
    int gpu = -1;  // HIP device the verdict is about; -1 = the report's GPU
    std::string pci_bdf;
  };
  std::vector<Verdict> verdicts;
  void verdict(const std::string& test, uint32_t system, int health, double value, const std::string& message, int for_gpu = -1) {
    if (health > GM_HEALTH_PASS) complain(health, for_gpu >= 0 ? "GPU " + std::to_string(for_gpu) + ": " + message : message);
    for (auto& v : verdicts)
      if (v.test == test && v.gpu == for_gpu) {
        if (health > v.health) v.health = health, v.value = value, v.message = message;
        return;
      }
    Verdict v{test, system, health, value, message};
    v.gpu = for_gpu;
    verdicts.push_back(v);
  }
  void verdict_for_gpu(int g, const std::string& test, uint32_t system, int health, double value, const std::string& message) {
    verdict(test, system, health, value, message, g);
  }
  void verdict_fail(const std::string& test, uint32_t system, double value, const char* fmt, ...) __attribute__((format(printf, 5, 6))) {
    char buf[192];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    verdict(test, system, GM_HEALTH_FAIL, value, buf);
  }
  void verdict_pass(const std::string& test, uint32_t system, double value, const std::string& message) {
    verdict(test, system, GM_HEALTH_PASS, value, message);
  }
  // A throughput against its expectation: records WARN / FAIL, returns the health.
  int grade(const std::string& test, uint32_t system, const std::string& what, double value, double want, const diag_expect::Expect& e,
            const char* unit) {
    const int h = diag_expect::grade_rate(value, want, e);
    if (h) {
      char b[192];
      snprintf(b, sizeof b, "%s %.4g %s is %.0f %% of the %.4g %s expected", what.c_str(), value, unit, 100 * value / want, want, unit);
      verdict(test, system, h, value, b);
    }
    return h;
  }

  static std::string escape(const std::string& in) {
    std::string o;
    for (char c : in) {
      if (c == '"' || c == '\\') o += '\\';
      o += (unsigned char)c < 0x20 ? ' ' : c;
    }
    return o;
  }
  std::string verdicts_json(const std::function<std::string(int)>& bdf_of) const {
    std::string vj = "[";
    for (size_t i = 0; i < verdicts.size(); i++) {
      const Verdict& v = verdicts[i];
      char num[64];
      if (std::isfinite(v.value)) snprintf(num, sizeof num, "%.6g", v.value); else snprintf(num, sizeof num, "null");
      const int g = v.gpu >= 0 ? v.gpu : gpu;
      const std::string bdf = !v.pci_bdf.empty() ? v.pci_bdf : g >= 0 ? bdf_of(g) : "";
      vj += std::string(i ? "," : "") + "{\"test\":\"" + v.test + "\"," + (g >= 0 ? "\"gpu\":" + std::to_string(g) + "," : "") +
            (bdf.empty() ? "" : "\"pci_bdf\":\"" + escape(bdf) + "\",") + "\"system\":" + std::to_string(v.system) + ",\"health\":" +
            std::to_string(v.health) + ",\"value\":" + num + ",\"message\":\"" + escape(v.message) + "\"}";
    }
    return vj + "]";
  }
  const char* result() const { return failed ? "fail" : warned ? "warn" : "pass"; }
  // The whole report as one JSON object; `gpu`, `pci_bdf` and `timestamp` let the exporter label and age it.  A node
  // report carries its per-GPU reports under "gpus" and every verdict (its own and theirs) under "verdicts".
  std::string document(const std::function<std::string(int)>& bdf_of) const {
    std::string d = "{\"scope\":\"" + std::string(gpu >= 0 ? "gpu" : "node") + "\"";
    if (gpu >= 0) d += ",\"gpu\":" + std::to_string(gpu) + ",\"pci_bdf\":\"" + escape(pci_bdf.empty() ? bdf_of(gpu) : pci_bdf) + "\"";
    d += ",\"timestamp\":" + std::to_string((long long)time(nullptr));
    if (!js.empty()) d += "," + js;
    if (!subs.empty()) {
      d += ",\"gpus\":[";
      for (size_t i = 0; i < subs.size(); i++) d += (i ? "," : "") + subs[i].document(bdf_of);
      d += "]";
    }
    Report all = all_verdicts();
    return d + ",\"verdicts\":" + all.verdicts_json(bdf_of) + ",\"result\":\"" + result() + "\"}";
  }
  // This report's verdicts and, for a node report, those of its per-GPU reports with the GPU filled in.
  Report all_verdicts() const {
    Report all;
    all.gpu = gpu;
    all.verdicts = verdicts;
    for (auto& s : subs)
      for (auto v : s.verdicts) {
        if (v.gpu < 0) v.gpu = s.gpu;
        all.verdicts.push_back(v);
      }
    return all;
  }
  int finish(const std::string& publish_path, const std::function<std::string(int)>& bdf_of) {
    const std::string doc = document(bdf_of);
    if (json) printf("%s\n", doc.c_str());
    else printf("\nResult: %s\n", failed ? "FAIL" : warned ? "WARN" : "PASS");
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

// PCI address of a HIP device ("" when it cannot be read): the identity that the engine, whose numbering follows
// the PCI bus, and the compute runtime, whose numbering may not, share.
std::string bdf_of_hip(int gpu) {
  static std::mutex mu;
  static std::map<int, std::string> cache;
  std::lock_guard<std::mutex> l(mu);
  auto it = cache.find(gpu);
  if (it != cache.end()) return it->second;
  gmd_device_info_t d;
  std::string b = gmd_device_info(gpu, &d) == GMD_OK ? d.pci_bdf : "";
  cache[gpu] = b;
  return b;
}

// The expectation table entry of a GPU with the command line's replacements applied.
bool expect_for(const Args& a, int gpu, diag_expect::Expect* e) {
  gmd_device_info_t d;
  memset(&d, 0, sizeof d);
  if (!a.has("--no-expect") && gmd_device_info(gpu, &d) == GMD_OK) *e = diag_expect::lookup(d.gcn_arch, d.name);
  std::string bad;
  if (a.has("--expect") && !a.has("--no-expect") && !diag_expect::apply_overrides(a.str("--expect"), e, &bad)) {
    fprintf(stderr, "--expect: cannot read '%s' (key=value, value >= 0)\n", bad.c_str());
    return false;
  }
  e->scale = a.num("--expect-scale", 1.0);
  if (!(e->scale > 0)) e->scale = 1.0;
  return true;
}

// -g N, --gpus a,b,c or --gpus all.  Empty = no GPU visible or a bad list.
std::vector<int> target_gpus(const Args& a) {
  std::vector<int> ids;
  const int visible = gmd_device_count();
  if (a.has("--gpus")) {
    if (a.str("--gpus") == "all")
      for (int i = 0; i < visible; i++) ids.push_back(i);
    else
      for (auto& t : csv(a.str("--gpus"))) {
        const int g = atoi(t.c_str());
        // GPUMON_DIAG_ALLOW_REPEAT=1 lets a GPU appear twice: two host threads on one card, which is how the
        // concurrent path is exercised on a machine with a single GPU
        const bool repeat_ok = getenv("GPUMON_DIAG_ALLOW_REPEAT") != nullptr;
        if (g < 0 || g >= visible || (!repeat_ok && std::find(ids.begin(), ids.end(), g) != ids.end())) {
          fprintf(stderr, "--gpus: bad or repeated GPU %s (%d visible)\n", t.c_str(), visible);
          return {};
        }
        ids.push_back(g);
      }
  } else {
    ids.push_back((int)a.num("-g", 0));
  }
  return ids;
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

int cmd_hbm(const Args& a, Report& r, int gpu) {
  const uint64_t bytes = a.size("--bytes", 4ull << 30);
  const int iters = (int)a.num("--iters", 10);
  const std::string which = a.str("--kernel", "all");
  const double floor_tbps = a.num("--min-tbps", 0);
  diag_expect::Expect e;
  if (!expect_for(a, gpu, &e)) return 2;
  r.text("HBM stream on GPU %d, %.1f GiB per buffer, %d iterations\n  %-8s %10s %10s %10s %8s\n", gpu, bytes / 1073741824.0, iters,
         "kernel", "best TB/s", "median", "expected", "check");
  std::string obj = "{";
  bool any = false;
  double worst = 1e30;
  for (int k = 0; k < GMD_STREAM_KERNELS; k++) {
    if (which != "all" && which != kStreamNames[k]) continue;
    gmd_stream_result_t s;
    DIAG_CALL(r, gmd_hbm_stream(gpu, k, bytes, iters, 2, &s));
    const std::string key = std::string("hbm.") + kStreamNames[k];
    // the table's rates hold for buffers far beyond the 256 MiB Infinity Cache
    const double want = bytes >= (1ull << 30) ? e.want(key) : 0;
    r.text("  %-8s %10.3f %10.3f %10s %8s\n", kStreamNames[k], s.tbps_best, s.tbps_median, want > 0 ? strprintf_local("%.2f", want).c_str() : "-",
           s.verified ? "ok" : "WRONG");
    obj += std::string(any ? "," : "") + "\"" + kStreamNames[k] + "\":{\"tbps_best\":" + jnum(s.tbps_best) +
           ",\"tbps_median\":" + jnum(s.tbps_median) + ",\"expected_tbps\":" + (want > 0 ? jnum(want) : "null") + ",\"verified\":" +
           (s.verified ? "true" : "false") + "}";
    any = true;
    r.metrics[key] = s.tbps_best;
    if (!s.verified) r.verdict_fail("hbm_stream", GM_HEALTH_WATCH_MEM, s.tbps_best, "stream %s produced wrong data", kStreamNames[k]);
    if (floor_tbps > 0 && s.tbps_best < floor_tbps)
      r.verdict_fail("hbm_stream", GM_HEALTH_WATCH_MEM, s.tbps_best, "stream %s %.3f TB/s is under the %.3f TB/s floor", kStreamNames[k], s.tbps_best, floor_tbps);
    else if (floor_tbps <= 0)
      r.grade("hbm_stream", GM_HEALTH_WATCH_MEM, std::string("stream ") + kStreamNames[k], s.tbps_best, want, e, "TB/s");
    worst = std::min(worst, s.tbps_best);
  }
  if (!any) {
    fprintf(stderr, "unknown kernel %s\n", which.c_str());
    return 2;
  }
  r.verdict_pass("hbm_stream", GM_HEALTH_WATCH_MEM, worst, e.known ? "every stream kernel verified and at its expected rate" : "every stream kernel verified");
  r.member("hbm_stream", obj + "}");
  return 0;
}

// The negotiated PCIe link of a GPU from sysfs ("32.0 GT/s PCIe", "16"); flags override, 0 = unknown.
void pcie_link_of(const Args& a, int gpu, int* gen, int* width) {
  *gen = (int)a.num("--pcie-gen", 0), *width = (int)a.num("--pcie-width", 0);
  const std::string dir = "/sys/bus/pci/devices/" + bdf_of_hip(gpu) + "/";
  auto first_line = [](const std::string& path) {
    std::string s;
    if (FILE* f = fopen(path.c_str(), "r")) {
      char buf[128];
      if (fgets(buf, sizeof buf, f)) s = buf;
      fclose(f);
    }
    return s;
  };
  if (*gen <= 0) {
    const double gts = atof(first_line(dir + "current_link_speed").c_str());
    *gen = gts >= 63 ? 6 : gts >= 31 ? 5 : gts >= 15 ? 4 : gts >= 7.5 ? 3 : gts >= 4.5 ? 2 : gts >= 2 ? 1 : 0;
  }
  if (*width <= 0) *width = atoi(first_line(dir + "current_link_width").c_str());
}

int cmd_pcie(const Args& a, Report& r, int gpu) {
  diag_expect::Expect e;
  if (!expect_for(a, gpu, &e)) return 2;
  gmd_pcie_config_t c;
  memset(&c, 0, sizeof c);
  c.bytes = a.size("--bytes", 256ull << 20);
  c.iters = (int)a.num("--iters", 5);
  c.verify = a.has("--no-verify") ? 0 : 1;
  c.workgroups = (int)a.num("--workgroups", 0);
  gmd_pcie_result_t p;
  {
    const int prc = gmd_pcie_test(gpu, &c, &p);
    if (prc == GMD_ERR_NOMEM || (prc == GMD_ERR_HIP && strstr(gmd_last_error(), "hipHostMalloc"))) {
      // Half a gigabyte of pinned host memory: a job whose memory limit does not allow it cannot run this test.  (The
      // driver's pinning is not subject to RLIMIT_MEMLOCK: `ulimit -l 64` does not stop it.)  That is a fact about
      // the job's environment, not about the link.
      r.text("host link of GPU %d: not tested, the host memory could not be pinned (%s)\n", gpu, gmd_last_error());
      r.member("pcie", "{\"skipped\":\"cannot pin host memory\"}");
      return 0;
    }
    if (prc != GMD_OK) {
      fprintf(stderr, "error: gmd_pcie_test -> %d: %s\n", prc, gmd_last_error());
      return 2;
    }
  }
  int gen = 0, width = 0;
  pcie_link_of(a, gpu, &gen, &width);
  const double link = gmd_pcie_link_gbps(gen, width);
  r.text("host link of GPU %d: %s, %.0f MiB per transfer, %d iterations\n", gpu,
         link > 0 ? strprintf_local("PCIe generation %d x%d, %.1f GB/s per direction", gen, width, link).c_str() : "link generation and width unknown", p.bytes / 1048576.0, p.iters);
  const struct { const char* key; const char* what; double got; uint64_t bad; } rows[5] = {
      {"h2d", "copy engine, host to device", p.h2d_gbps, p.bad_h2d},
      {"d2h", "copy engine, device to host", p.d2h_gbps, p.bad_d2h},
      {"kernel_read", "kernel reads host memory", p.kernel_read_gbps, p.bad_kernel_read},
      {"kernel_write", "kernel writes host memory", p.kernel_write_gbps, p.bad_kernel_write},
      // Both directions busy: by the copy engines or by the two kernels, whichever pair got more through.  Whether two
      // copy-engine transfers overlap is the runtime's choice of engines (on some hosts they run one after the other),
      // and two kernels share the chip's path to the link; the better of the two is what the link is known to carry.
      {"bidir", "both directions at once (slower)", std::max(std::min(p.duplex_read_gbps, p.duplex_write_gbps), std::min(p.bidir_h2d_gbps, p.bidir_d2h_gbps)), 0}};
  std::string obj = "{\"link_gen\":" + std::to_string(gen) + ",\"link_width\":" + std::to_string(width) + ",\"link_gbps\":" + jnum(link) +
                    ",\"bytes\":" + std::to_string(p.bytes) + ",\"copy_bidir_h2d_gbps\":" + jnum(p.bidir_h2d_gbps) + ",\"copy_bidir_d2h_gbps\":" + jnum(p.bidir_d2h_gbps) +
                    ",\"duplex_read_gbps\":" + jnum(p.duplex_read_gbps) + ",\"duplex_write_gbps\":" + jnum(p.duplex_write_gbps) +
                    ",\"small_h2d_us\":" + jnum(p.small_h2d_us) + ",\"small_d2h_us\":" + jnum(p.small_d2h_us);
  r.text("  %-34s %9s %9s %8s\n", "traffic", "GB/s", "expected", "check");
  uint64_t bad_total = 0;
  double worst_share = 1e30;
  for (auto& row : rows) {
    const double want = link > 0 ? e.want(std::string("pcie.") + row.key) * link : 0;
    r.text("  %-34s %9.1f %9s %8s\n", row.what, row.got, want > 0 ? strprintf_local("%.1f", want).c_str() : "-",
           !p.verified ? "-" : row.bad ? "WRONG" : "ok");
    obj += std::string(",\"") + row.key + "_gbps\":" + jnum(row.got) + ",\"" + row.key + "_expected_gbps\":" + (want > 0 ? jnum(want) : "null");
    r.metrics[std::string("pcie.") + row.key] = row.got;
    bad_total += row.bad;
    if (row.bad) r.verdict_fail("pcie_link", GM_HEALTH_WATCH_PCIE, (double)row.bad, "%s: %llu corrupt 16-byte words", row.what, (unsigned long long)row.bad);
    r.grade("pcie_link", GM_HEALTH_WATCH_PCIE, row.what, row.got, want, e, "GB/s");
    if (link > 0) worst_share = std::min(worst_share, row.got / link);
  }
  const bool serial = p.bidir_d2h_gbps > 0 && p.bidir_h2d_gbps + p.bidir_d2h_gbps < 1.25 * std::max(p.h2d_gbps, p.d2h_gbps);
  r.text("  both kernels at once: %.1f read + %.1f written GB/s; both copy engines at once: %.1f up + %.1f down%s\n"
         "  an 8-byte copy takes %.1f us up, %.1f us down\n",
         p.duplex_read_gbps, p.duplex_write_gbps, p.bidir_h2d_gbps, p.bidir_d2h_gbps,
         serial ? " (the runtime ran them one after the other on this host)" : "", p.small_h2d_us, p.small_d2h_us);
  obj += ",\"bad_words\":" + std::to_string(bad_total) + "}";
  r.member("pcie", obj);
  r.verdict_pass("pcie_link", GM_HEALTH_WATCH_PCIE, link > 0 ? 100 * worst_share : p.h2d_gbps,
                 link > 0 && e.known ? "every transfer verified and at its expected share of the link" : "every transfer verified");
  return 0;
}

void progress_cb(double f, const char* what, void* user) {
  if (*(bool*)user) return;
  fprintf(stderr, "\r  %5.1f %%  %-24s", f * 100, what);
  if (f >= 1.0) fprintf(stderr, "\n");
}

int cmd_memtest(const Args& a, Report& r, int gpu) {
  diag_expect::Expect e;
  if (!expect_for(a, gpu, &e)) return 2;
  gmd_memtest_config_t c;
  memset(&c, 0, sizeof c);
  c.bytes = a.size("--bytes", 0);
  c.fraction = a.num("--fraction", 0.90);
  c.chunk_bytes = a.size("--chunk", 0);
  c.passes = (uint32_t)a.num("--passes", 1);
  c.march = a.has("--march") ? 1 : 0;
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
  bool quiet = r.json || r.buffered || !isatty(2);
  c.progress = progress_cb;
  c.progress_user = &quiet;
  gmd_memtest_result_t m;
  DIAG_CALL(r, gmd_memtest(gpu, &c, &m));
  r.text("memtest on GPU %d: %.1f GiB in %u chunks, %u patterns, %.2f s\n  write %.2f TB/s  verify %.2f TB/s  errors %llu%s\n", gpu,
         m.bytes_tested / 1073741824.0, m.chunks, m.patterns_run, m.seconds, m.write_tbps, m.verify_tbps,
         (unsigned long long)m.errors, m.scrubbed ? "" : "  (no memory left for the cache scrub: the last lines written may have been checked in the cache)");
  if (m.march_passes) r.text("  march (check + write complement, up and down): %.2f TB/s read + written\n", m.march_tbps);
  std::string log = "[";
  for (uint32_t i = 0; i < m.logged; i++) {
    r.text("  offset 0x%llx (chunk %u, address 0x%llx, +0x%03x in its 4 KiB page) pattern 0x%x expected %016llx got %016llx: %u bit%s\n",
           (unsigned long long)m.log[i].byte_offset, m.log[i].chunk, (unsigned long long)m.log[i].device_address, m.log[i].page_offset,
           m.log[i].pattern, (unsigned long long)m.log[i].expected, (unsigned long long)m.log[i].actual, m.log[i].flipped_bits,
           m.log[i].flipped_bits == 1 ? "" : "s");
    log += std::string(i ? "," : "") + "{\"offset\":" + std::to_string(m.log[i].byte_offset) + ",\"pattern\":" +
           std::to_string(m.log[i].pattern) + ",\"chunk\":" + std::to_string(m.log[i].chunk) + ",\"device_address\":" +
           std::to_string(m.log[i].device_address) + ",\"page_offset\":" + std::to_string(m.log[i].page_offset) + ",\"flipped_bits\":" +
           std::to_string(m.log[i].flipped_bits) + "}";
  }
  r.member("memtest", "{\"bytes_tested\":" + std::to_string(m.bytes_tested) + ",\"errors\":" + std::to_string(m.errors) +
                          ",\"patterns\":" + std::to_string(m.patterns_run) + ",\"write_tbps\":" + jnum(m.write_tbps) +
                          ",\"verify_tbps\":" + jnum(m.verify_tbps) + ",\"march_passes\":" + std::to_string(m.march_passes) + ",\"march_tbps\":" +
                          jnum(m.march_tbps) + ",\"scrubbed\":" + (m.scrubbed ? "true" : "false") + ",\"seconds\":" + jnum(m.seconds) +
                          ",\"log\":" + log + "]}");
  if (m.errors) r.verdict_fail("memtest", GM_HEALTH_WATCH_MEM, (double)m.errors, "memtest found %llu bad words in %.1f GiB", (unsigned long long)m.errors, m.bytes_tested / 1073741824.0);
  // rates: only where the region is large enough for launch costs and the cache not to decide
  if (m.bytes_tested >= (4ull << 30)) {
    r.metrics["memtest.write"] = m.write_tbps, r.metrics["memtest.verify"] = m.verify_tbps;
    r.grade("memtest", GM_HEALTH_WATCH_MEM, "memtest write pass", m.write_tbps, e.want("memtest.write"), e, "TB/s");
    r.grade("memtest", GM_HEALTH_WATCH_MEM, "memtest verify pass", m.verify_tbps, e.want("memtest.verify"), e, "TB/s");
    if (m.march_passes) {
      r.metrics["memtest.march"] = m.march_tbps;
      r.grade("memtest", GM_HEALTH_WATCH_MEM, "memtest march", m.march_tbps, e.want("memtest.march"), e, "TB/s");
    }
  }
  r.verdict_pass("memtest", GM_HEALTH_WATCH_MEM, 0, "no bad words");
  return 0;
}

int cmd_mfma(const Args& a, Report& r, int gpu) {
  diag_expect::Expect e;
  if (!expect_for(a, gpu, &e)) return 2;
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
    r.metrics[std::string("mfma.") + kDtypeNames[dt]] = m.tflops;
    r.grade("mfma_peak", GM_HEALTH_WATCH_COMPUTE, std::string("matrix-pipe peak ") + kDtypeNames[dt], m.tflops, e.want(std::string("mfma.") + kDtypeNames[dt]), e, "TFLOP/s");
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
  void stop() {
    if (pid < 0) return;
    kill(pid, SIGTERM);
    int st = 0;
    waitpid(pid, &st, 0);
    pid = -1;
    if (reader.joinable()) reader.join();
  }
  ~MfmaWatch() { stop(); }
  std::string fold(Report& r, const char* pci_bdf) {
    std::lock_guard<std::mutex> l(mu);
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

// Engine-side telemetry of the GPU under stress: power / temperature / clocks at 100 ms, and what the soak is
// judged by: the clock it held, how hot it got, whether it could draw its power, whether it throttled thermally.
struct Soak {
  bool on = false;
  uint32_t gpu = 0;
  int64_t t0 = 0;
  double power_cap_w = NAN;
  gm_group_t group = 0;
  gm_field_group_t fg = 0;
  struct Stat {
    double lo = 1e300, hi = -1e300, sum = 0, first = NAN, last = NAN;
    int n = 0;
    double avg() const { return n ? sum / n : NAN; }
  };
  std::map<std::string, Stat> stats;
  bool start(int hip_dev) {
    const std::string bdf = bdf_of_hip(hip_dev);
    if (bdf.empty()) return false;
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
      if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, bdf.c_str()) == 0) gpu = i, found = true, power_cap_w = at.power_cap_w;
    }
    if (!found) {
      gm_shutdown();
      return false;
    }
    uint16_t ids[] = {GM_FI_POWER_USAGE, GM_FI_GPU_TEMP, GM_FI_MEM_TEMP, GM_FI_GFX_CLOCK, GM_FI_MEM_CLOCK, GM_FI_GFX_UTIL,
                      GM_FI_THERMAL_VIOLATION, GM_FI_POWER_VIOLATION};
    const std::string name = "diag-soak-" + std::to_string(hip_dev);
    gm_group_create(name.c_str(), &group);
    gm_group_add_device(group, gpu);
    gm_field_group_create(ids, 8, name.c_str(), &fg);
    gm_watch_fields(group, fg, 100000, 7200.0, 0);
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
      bool counter;
    } fs[] = {{GM_FI_POWER_USAGE, "power", "W", false},     {GM_FI_GPU_TEMP, "gpu_temp", "C", false},   {GM_FI_MEM_TEMP, "hbm_temp", "C", false},
              {GM_FI_GFX_CLOCK, "gfx_clock", "MHz", false}, {GM_FI_MEM_CLOCK, "mem_clock", "MHz", false}, {GM_FI_GFX_UTIL, "gfx_util", "%", false},
              {GM_FI_THERMAL_VIOLATION, "thermal_throttle_us", "us", true}, {GM_FI_POWER_VIOLATION, "power_throttle_us", "us", true}};
    std::string obj = "{";
    r.text("  telemetry during the run (100 ms samples):\n");
    bool first = true;
    for (auto& f : fs) {
      std::vector<gm_value_t> v(65536);  // 100 ms samples: close to two hours
      int n = (int)v.size();
      if (gm_values_since(gpu, f.id, t0, v.data(), &n) != GM_ST_OK) continue;
      Stat st;
      for (int i = 0; i < n; i++) {
        if (v[i].status != GM_VS_OK) continue;
        double x = v[i].type == GM_FT_DOUBLE ? v[i].v.dbl : (double)v[i].v.i64;
        if (!st.n) st.first = x;
        st.last = x;
        st.lo = std::min(st.lo, x), st.hi = std::max(st.hi, x), st.sum += x, st.n++;
      }
      if (!st.n) continue;
      stats[f.name] = st;
      if (f.counter) {
        r.text("    %-20s grew by %.0f %s  (%d samples)\n", f.name, st.last - st.first, f.unit, st.n);
        obj += std::string(first ? "" : ",") + "\"" + f.name + "\":{\"delta\":" + jnum(st.last - st.first) + ",\"samples\":" + std::to_string(st.n) + "}";
      } else {
        r.text("    %-10s min %8.1f  avg %8.1f  max %8.1f %s  (%d samples)\n", f.name, st.lo, st.avg(), st.hi, f.unit, st.n);
        obj += std::string(first ? "" : ",") + "\"" + f.name + "\":{\"min\":" + jnum(st.lo) + ",\"avg\":" + jnum(st.avg()) +
               ",\"max\":" + jnum(st.hi) + ",\"samples\":" + std::to_string(st.n) + "}";
      }
      first = false;
    }
    if (std::isfinite(power_cap_w)) obj += std::string(first ? "" : ",") + "\"power_cap\":" + jnum(power_cap_w);
    return obj + "}";
  }
  // The soak's verdicts (after summarize).  `seconds` = how long the GPU was under load.
  void judge(Report& r, const diag_expect::Expect& e, double seconds) {
    if (!on) return;
    auto has = [&](const char* k) { return stats.count(k) && stats[k].n > 0; };
    if (has("gfx_clock")) r.metrics["soak.gfx_clock_mhz"] = stats["gfx_clock"].avg();
    if (has("power")) r.metrics["soak.power_w"] = stats["power"].avg();
    if (!e.known || seconds < 2) return;
    if (has("gfx_clock")) {
      const double clk = stats["gfx_clock"].avg();
      r.grade("soak_clock", GM_HEALTH_WATCH_POWER, "shader clock held under load", clk, e.want("soak.gfx_clock_mhz"), e, "MHz");
      r.verdict_pass("soak_clock", GM_HEALTH_WATCH_POWER, clk, strprintf_local("held %.0f MHz under load", clk));
    }
    if (has("power")) {
      if (std::isfinite(power_cap_w) && power_cap_w > 0) {
        // a healthy part under a matrix load comes close to its cap; one that cannot is held back by something else
        const double frac = stats["power"].hi / power_cap_w;
        const int h = diag_expect::grade_rate(frac, e.get("soak.power_fraction"), e);
        r.verdict("soak_power", GM_HEALTH_WATCH_POWER, h, stats["power"].hi,
                  strprintf_local("drew at most %.0f W, %.0f %% of the %.0f W cap%s", stats["power"].hi, 100 * frac, power_cap_w,
                                  h ? strprintf_local(" (expected %.0f %%)", 100 * e.get("soak.power_fraction")).c_str() : ""));
      }
    }
    int h_thermal = GM_HEALTH_PASS;
    std::string why = "within limits";
    if (has("gpu_temp")) {
      const int h = diag_expect::grade_limit(stats["gpu_temp"].hi, e.get("limit.hotspot_c.warn"), e.get("limit.hotspot_c.fail"));
      if (h > h_thermal) h_thermal = h, why = strprintf_local("hotspot reached %.0f C (warn at %.0f, fail at %.0f)", stats["gpu_temp"].hi, e.get("limit.hotspot_c.warn"), e.get("limit.hotspot_c.fail"));
    }
    if (has("hbm_temp")) {
      const int h = diag_expect::grade_limit(stats["hbm_temp"].hi, e.get("limit.hbm_c.warn"), e.get("limit.hbm_c.fail"));
      if (h > h_thermal) h_thermal = h, why = strprintf_local("HBM reached %.0f C (warn at %.0f, fail at %.0f)", stats["hbm_temp"].hi, e.get("limit.hbm_c.warn"), e.get("limit.hbm_c.fail"));
    }
    if (has("thermal_throttle_us")) {
      const double frac = (stats["thermal_throttle_us"].last - stats["thermal_throttle_us"].first) / (seconds * 1e6);
      const int h = diag_expect::grade_limit(frac, e.get("limit.thermal_residency.warn"), e.get("limit.thermal_residency.fail"));
      if (h > h_thermal) h_thermal = h, why = strprintf_local("thermally throttled for %.1f %% of the soak", 100 * frac);
    }
    if (has("gpu_temp") || has("hbm_temp") || has("thermal_throttle_us"))
      r.verdict("soak_thermal", GM_HEALTH_WATCH_THERMAL, h_thermal, has("gpu_temp") ? stats["gpu_temp"].hi : NAN, why);
  }
  void stop() {
    if (on) gm_shutdown();
    on = false;
  }
};

int cmd_gemm(const Args& a, Report& r, int gpu, MfmaWatch* mfma) {
  const std::string which = a.str("--dtype", "bf16,mxfp8");
  const int m = (int)a.num("--m", 8192), n = (int)a.num("--n", 8192), k = (int)a.num("--k", 8192);
  const double secs = a.num("--seconds", 5.0), floor_tf = a.num("--min-tflops", 0);
  diag_expect::Expect e;
  if (!expect_for(a, gpu, &e)) return 2;
  const bool table_shape = m == 8192 && n == 8192 && k == 8192;  // the shape the table's rates are for
  Soak soak;
  if (!a.has("--no-telemetry") && !soak.start(gpu)) r.text("  note: engine telemetry unavailable; running without it\n");
  std::string obj = "{";
  bool any = false;
  double loaded_s = 0;
  const std::string bdf = bdf_of_hip(gpu);
  for (auto& name : csv(which)) {
    int dt = dtype_of(name);
    if (dt < 0 || dt == GMD_DT_F32) {
      fprintf(stderr, "unknown GEMM dtype %s\n", name.c_str());
      soak.stop();
      return 2;
    }
    gmd_gemm_stress_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.device = gpu, cfg.dtype = dt, cfg.m = m, cfg.n = n, cfg.k = k, cfg.seconds = secs, cfg.verify = 1;
    cfg.check_every = (int)a.num("--check-every", 0);
    gmd_mfma_result_t g;
    const double t_begin = MfmaWatch::now();
    // A long soak says every half minute that it is alive and what the part is doing (stderr: the report stays clean).
    std::atomic<bool> beat_stop{false};
    std::thread beat;
    if (secs >= 45)
      beat = std::thread([&] {
        for (int t = 30;; t += 30) {
          for (int i = 0; i < 300 && !beat_stop; i++) usleep(100000);
          if (beat_stop) return;
          std::string now;
          if (soak.on) {
            const uint16_t ids[3] = {GM_FI_POWER_USAGE, GM_FI_GFX_CLOCK, GM_FI_GPU_TEMP};
            gm_value_t v[3];
            if (gm_latest_values(soak.gpu, ids, 3, v) == GM_ST_OK) {
              auto num = [](const gm_value_t& x) { return x.status != GM_VS_OK ? NAN : x.type == GM_FT_DOUBLE ? x.v.dbl : (double)x.v.i64; };
              now = strprintf_local(": %.0f W, %.0f MHz, %.0f C", num(v[0]), num(v[1]), num(v[2]));
            }
          }
          fprintf(stderr, "  %s soak on GPU %d, %d of %.0f s%s\n", name.c_str(), gpu, t, secs, now.c_str());
        }
      });
    int rc = gmd_gemm_stress_run(&cfg, &g);
    beat_stop = true;
    if (beat.joinable()) beat.join();
    const double t_end = MfmaWatch::now();
    loaded_s += t_end - t_begin;
    if (rc != GMD_OK && rc != GMD_ERR_VERIFY) {
      fprintf(stderr, "error: gemm stress -> %d: %s\n", rc, gmd_last_error());
      soak.stop();
      return 2;
    }
    const std::string key = "gemm." + name;
    const double want = table_shape ? e.want(key) : 0;
    r.text("GEMM stress %s %dx%dx%d on GPU %d: %.1f TFLOP/s (best %.1f%s), %llu launches in %.1f s, sdc mismatches %llu, "
           "max rel err vs fp32 reference %.2e, check %s\n",
           name.c_str(), m, n, k, gpu, g.tflops, g.tflops_best, want > 0 ? strprintf_local(", expected %.0f", want).c_str() : "",
           (unsigned long long)g.launches, g.seconds, (unsigned long long)g.sdc_mismatches, g.max_rel_err,
           g.verified > 0 ? "ok" : g.verified < 0 ? "WRONG" : "skipped");
    int schedule = 0, ranges = 1;
    if (gmd_gemm_plan(dt, m, n, k, &schedule, &ranges) == GMD_OK && schedule)
      r.text("  schedule: split along K into %d ranges on %s tiles, partial sums added up in a fixed order\n", ranges,
             schedule == 10 ? "256x256" : "128x128");
    if (g.verified)
      r.text("  integrity: %llu launches summed in the kernel's epilogue and held against the first, %llu re-read from memory\n",
             (unsigned long long)g.checked_in_kernel, (unsigned long long)g.checked_in_memory);
    obj += std::string(any ? "," : "") + "\"" + name + "\":{\"tflops\":" + jnum(g.tflops) + ",\"tflops_best\":" +
           jnum(g.tflops_best) + ",\"expected_tflops\":" + (want > 0 ? jnum(want) : "null") + ",\"launches\":" + std::to_string(g.launches) +
           ",\"sdc_mismatches\":" + std::to_string(g.sdc_mismatches) + ",\"max_rel_err\":" + jnum(g.max_rel_err) + ",\"verified\":" +
           std::to_string(g.verified) + ",\"checked_in_kernel\":" + std::to_string(g.checked_in_kernel) + ",\"checked_in_memory\":" +
           std::to_string(g.checked_in_memory) + ",\"split_k\":" + std::to_string(ranges);
    if (mfma) {  // the windows that lie inside this dtype's soak
      usleep(300000);  // let the window that ends with the soak arrive
      int wn = 0;
      const double avg = mfma->average_between(bdf.c_str(), t_begin, t_end, &wn);
      if (wn > 0) {
        r.text("  matrix pipes during the %s soak: %.1f %% busy (MFMA busy counter, %d windows)\n", name.c_str(), avg, wn);
        obj += ",\"mfma_active_avg\":" + jnum(avg);
      }
    }
    obj += "}";
    any = true;
    r.metrics[key] = g.tflops;
    const std::string test = "gemm_" + name;
    if (g.verified < 0 && !g.sdc_mismatches)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, g.max_rel_err,
                     "GEMM %s: sampled entries differ from the fp32 reference (max relative error %.3g, tolerance 4e-3)", name.c_str(), g.max_rel_err);
    if (g.sdc_mismatches)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, (double)g.sdc_mismatches,
                     "GEMM %s: %llu launches changed their checksum (silent data corruption)", name.c_str(), (unsigned long long)g.sdc_mismatches);
    if (floor_tf > 0 && g.tflops < floor_tf)
      r.verdict_fail(test, GM_HEALTH_WATCH_COMPUTE, g.tflops, "GEMM %s %.1f TFLOP/s is under the %.1f floor", name.c_str(), g.tflops, floor_tf);
    else if (floor_tf <= 0)
      r.grade(test, GM_HEALTH_WATCH_COMPUTE, "GEMM " + name, g.tflops, want, e, "TFLOP/s");
    r.verdict_pass(test, GM_HEALTH_WATCH_COMPUTE, g.tflops, strprintf_local("%.1f TFLOP/s, reference and checksums agree", g.tflops));
  }
  r.member("gemm_stress", obj + "}");
  r.member("telemetry", soak.summarize(r));
  soak.judge(r, e, loaded_s);
  soak.stop();
  return 0;
}

bool parse_gpus(const Args& a, std::vector<int>* ids) {
  if (!a.has("--gpus") || a.str("--gpus") == "all") return true;
  for (auto& s : csv(a.str("--gpus"))) ids->push_back(atoi(s.c_str()));
  return !ids->empty();
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
      const std::string bdf = bdf_of_hip(hip);
      for (uint32_t i = 0; i < n && idx < 0 && !bdf.empty(); i++) {
        gm_device_attributes_t at;
        memset(&at, 0, sizeof at);
        at.version = GM_DEVICE_ATTRIBUTES_VERSION;
        if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, bdf.c_str()) == 0) idx = (int)i;
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

// What the fabric of this node is expected to carry: the advertised rate of one direction of one link from the
// engine's link table (or --link-gbps, or the MI355X figure), the deratings of gmd_link_expect_t.
gmd_link_expect_t fabric_expect(const Args& a, const std::vector<int>& ids, LinkMap* map) {
  gmd_link_expect_t e;
  memset(&e, 0, sizeof e);
  if (map && !map->on) map->load(ids);
  if (a.has("--link-gbps")) e.link_gbps = a.num("--link-gbps", 0);
  else if (map && map->per_direction_gbytes > 0) e.link_gbps = map->per_direction_gbytes;
  e.warn_ratio = a.num("--warn-ratio", 0);
  e.fail_ratio = a.num("--fail-ratio", 0);
  e.scale = a.num("--expect-scale", 1.0);
  gmd_link_expect_defaults(&e);
  return e;
}

int run_sweep(const Args& a, Report& r, const std::vector<int>& ids, const gmd_link_expect_t& fe, std::string* js) {
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
  uint64_t peak_bytes[GMD_COLL_COUNT] = {0};
  int w = n ? pts[0].ndev : 0;
  r.text("# RCCL sweep on %d GPU%s (device-side timing; rows under about 1 MiB are bound by launch latency, not by the links)\n"
         "# %-15s%14s%7s%12s%13s%13s  check\n", w, w == 1 ? "" : "s", "op", "bytes", "iters", "time(us)", "algbw(GB/s)", "busbw(GB/s)");
  // gmd_rccl_sweep fails unless every communicator reports exactly the requested rank count on its own device
  *js += "{\"gpus\":" + std::to_string(w) + ",\"communicator_ranks\":" + std::to_string(w) + ",\"timing\":\"device events; rows under about 1 MiB are launch-latency bound\",\"points\":[";
  for (int i = 0; i < n; i++) {
    auto& p = pts[i];
    r.text("  %-15s%14llu%7d%12.1f%13.2f%13.2f  %s\n", gmd_collective_name(p.op), (unsigned long long)p.bytes, p.iters, p.time_us,
           p.algbw_gbps, p.busbw_gbps, p.verified > 0 ? "ok" : p.verified < 0 ? "MISMATCH" : "-");
    if (p.busbw_gbps > peak[p.op]) peak[p.op] = p.busbw_gbps, peak_bytes[p.op] = p.bytes;
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
  *js += "},\"expected_busbw_gbps\":{";
  // Expectations only mean something once links are involved (the 1-GPU row is a local copy) and once the sweep
  // reached sizes at which the links, not the launches, decide.
  const double floor_gbps = a.num("--min-busbw", 0);
  const bool graded = w > 1 && !a.has("--no-expect") && c.max_bytes >= (64ull << 20);
  first = true;
  for (int i = 0; i < GMD_COLL_COUNT; i++) {
    if (peak[i] <= 0 || w < 2) continue;
    const double want = floor_gbps > 0 ? floor_gbps : gmd_rccl_expected_busbw(&fe, i, w);
    *js += std::string(first ? "" : ",") + "\"" + gmd_collective_name(i) + "\":" + jnum(want);
    first = false;
    if (floor_gbps > 0) {
      if (peak[i] < floor_gbps)
        r.verdict_fail("rccl", GM_HEALTH_WATCH_XGMI, peak[i], "%s peaks at %.1f GB/s bus bandwidth on %d GPUs, under the %.1f GB/s floor",
                       gmd_collective_name(i), peak[i], w, floor_gbps);
    } else if (graded) {
      const int h = gmd_grade_against(peak[i], want, fe.warn_ratio, fe.fail_ratio);
      if (h)
        r.verdict("rccl", GM_HEALTH_WATCH_XGMI, h, peak[i],
                  strprintf_local("%s peaks at %.1f GB/s bus bandwidth on %d GPUs (at %llu bytes), %.0f %% of the %.1f GB/s expected of %.1f GB/s links",
                                  gmd_collective_name(i), peak[i], w, (unsigned long long)peak_bytes[i], 100 * peak[i] / want, want, fe.link_gbps));
    }
  }
  *js += "}}";
  r.verdict_pass("rccl", GM_HEALTH_WATCH_XGMI, NAN, graded || floor_gbps > 0 ? "every collective returned the right data at its expected bus bandwidth"
                                                                               : "every collective returned the right data");
  return 0;
}

int cmd_rccl(const Args& a, Report& r) {
  std::vector<int> ids;
  if (!parse_gpus(a, &ids)) return 2;
  int visible = gmd_device_count();
  if (ids.empty())
    for (int i = 0; i < visible; i++) ids.push_back(i);
  LinkMap map;
  const gmd_link_expect_t fe = ids.size() > 1 ? fabric_expect(a, ids, &map) : fabric_expect(a, ids, nullptr);
  if (!a.has("--scaling")) {
    std::string js;
    int rc = run_sweep(a, r, ids, fe, &js);
    if (rc) return rc;
    r.member("rccl", js);
    return 0;
  }
  // 1 / 2 / 4 / 8 ... GPUs: the scaling series.
  std::string arr = "[";
  for (size_t w = 1; w <= ids.size(); w *= 2) {
    std::vector<int> sub(ids.begin(), ids.begin() + w);
    std::string js;
    int rc = run_sweep(a, r, sub, fe, &js);
    if (rc) return rc;
    arr += std::string(w > 1 ? "," : "") + js;
  }
  r.member("rccl_scaling", arr + "]");
  return 0;
}

std::string matrix_json(const std::vector<double>& m, int w) {
  std::string mj = "[";
  for (int i = 0; i < w; i++) {
    mj += std::string(i ? "," : "") + "[";
    for (int j = 0; j < w; j++) mj += std::string(j ? "," : "") + jnum(m[(size_t)i * w + j]);
    mj += "]";
  }
  return mj + "]";
}

// xGMI link test: the peer-copy matrix in three traffic patterns, with remote stores and with remote loads,
// read together into findings that name the GPU, the link and the peer.
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
    // No link to test: prove the fused all-peers launch and its two clocks inside the one GPU instead.
    const int stand_ins = 7;
    std::vector<double> g(stand_ins);
    std::vector<uint64_t> bad(stand_ins);
    gmd_link_fan_result_t fr;
    DIAG_CALL(r, gmd_link_fan_selftest(ids[0], stand_ins, a.size("--bytes", 64ull << 20), (int)a.num("--workgroups", 0), (int)a.num("--iters", 5), g.data(),
                                       bad.data(), &fr));
    uint64_t nbad = 0;
    for (auto b : bad) nbad += b;
    const bool clocks_agree = fr.window_ms_max <= fr.launch_ms * 1.02 + 0.05 && fr.window_ms_min > 0;
    r.text("# xGMI link test: one GPU visible, no link to test\n# self-test of the all-peers launch on GPU %d: %d local destinations, %.1f GB/s in total, "
           "per-destination windows %.2f..%.2f ms inside a %.2f ms launch, %llu bad words\n",
           ids[0], stand_ins, fr.launch_gbps, fr.window_ms_min, fr.window_ms_max, fr.launch_ms, (unsigned long long)nbad);
    r.member("links", "{\"gpus\":1,\"tested\":0,\"selftest\":{\"destinations\":" + std::to_string(stand_ins) + ",\"launch_gbps\":" + jnum(fr.launch_gbps) +
                          ",\"launch_ms\":" + jnum(fr.launch_ms) + ",\"window_ms_min\":" + jnum(fr.window_ms_min) + ",\"window_ms_max\":" +
                          jnum(fr.window_ms_max) + ",\"bad_words\":" + std::to_string(nbad) + "}}");
    r.member("p2p_gbps", "[[0]]");
    if (nbad) r.verdict_fail("xgmi_p2p", GM_HEALTH_WATCH_XGMI, (double)nbad, "link-test self-test: %llu corrupt words inside GPU %d", (unsigned long long)nbad, ids[0]);
    if (!clocks_agree)
      r.verdict_fail("xgmi_p2p", GM_HEALTH_WATCH_XGMI, fr.window_ms_max, "link-test self-test: stamp windows (%.3f..%.3f ms) do not fit the %.3f ms launch",
                     fr.window_ms_min, fr.window_ms_max, fr.launch_ms);
    return 0;
  }
  LinkMap map;
  gmd_link_expect_t fe = fabric_expect(a, ids, &map);
  // --floor-gbps replaces the expectation of one direction of one link outright (and scales the others with it)
  if (a.has("--floor-gbps") && a.num("--floor-gbps", 0) > 0) fe.scale *= a.num("--floor-gbps", 0) / (fe.fail_ratio * gmd_link_expected_gbps(&fe, GMD_LINK_UNI, 1));

  gmd_link_config_t c;
  memset(&c, 0, sizeof c);
  c.ndev = w;
  c.devices = ids.data();
  c.bytes = a.size("--bytes", 256ull << 20);
  c.iters = (int)a.num("--iters", 5);
  c.verify = a.has("--no-verify") ? 0 : 1;
  c.workgroups = (int)a.num("--workgroups", 0);

  const char* names[] = {"uni", "bidi", "all"};
  const char* what[] = {"one pair at a time", "both directions of a pair at once", "every GPU on every peer at once"};
  std::vector<std::string> modes = a.has("--modes") ? csv(a.str("--modes")) : std::vector<std::string>{"uni", "bidi", "all"};
  const std::string direction = a.str("--direction", "both");
  if (direction != "push" && direction != "pull" && direction != "both") {
    fprintf(stderr, "--direction takes push, pull or both\n");
    return 2;
  }
  std::string js = "{\"gpus\":" + std::to_string(w) + ",\"link_gbps\":" + jnum(fe.link_gbps) + ",\"warn_ratio\":" + jnum(fe.warn_ratio) +
                   ",\"fail_ratio\":" + jnum(fe.fail_ratio);
  std::vector<int> worst(w, GM_HEALTH_PASS);
  std::vector<std::string> worst_msg(w);
  std::vector<double> slowest(w, 1e300);
  std::string findings = "[";
  int n_findings = 0;
  for (int pull = 0; pull < 2; pull++) {
    if ((pull && direction == "push") || (!pull && direction == "pull")) continue;
    c.pull = pull;
    fe.pull = pull;
    std::vector<double> mat[3], egress(w, 0);
    std::vector<uint64_t> bad_total((size_t)w * w, 0);
    js += std::string(",\"") + (pull ? "pull" : "push") + "\":{\"expected_gbps\":{";
    for (int mode = 0; mode < 3; mode++)
      js += std::string(mode ? "," : "") + "\"" + names[mode] + "\":" + jnum(gmd_link_expected_gbps(&fe, mode, 1));
    js += "}";
    for (int mode = 0; mode < 3; mode++) {
      // the one-at-a-time matrix is what the others are read against: it always runs
      if (mode != GMD_LINK_UNI && std::find(modes.begin(), modes.end(), names[mode]) == modes.end()) continue;
      std::vector<double>& m = mat[mode];
      m.assign((size_t)w * w, 0);
      std::vector<uint64_t> bad((size_t)w * w);
      c.per_gpu_gbps = mode == GMD_LINK_ALL ? egress.data() : nullptr;
      DIAG_CALL(r, gmd_link_test(&c, mode, m.data(), bad.data()));
      for (size_t x = 0; x < bad.size(); x++) bad_total[x] += bad[x];
      r.text("# link bandwidth, GB/s, %s, %s (row = GPU that runs the copy; expected %.1f per link)\n", pull ? "remote loads" : "remote stores", what[mode],
             gmd_link_expected_gbps(&fe, mode, 1));
      for (int i = 0; i < w; i++) {
        r.text("  ");
        for (int j = 0; j < w; j++) {
          r.text("%8.1f", m[(size_t)i * w + j]);
          if (i != j && m[(size_t)i * w + j] >= 0) slowest[i] = std::min(slowest[i], m[(size_t)i * w + j]);
        }
        if (mode == GMD_LINK_ALL) r.text("   | %8.1f by the launch's own clock", egress[i]);
        r.text("\n");
      }
      js += std::string(",\"") + names[mode] + "\":" + matrix_json(m, w);
      if (mode == GMD_LINK_ALL) {
        js += ",\"per_gpu_gbps\":[";
        for (int i = 0; i < w; i++) js += std::string(i ? "," : "") + jnum(egress[i]);
        js += "]";
      }
      if (mode == GMD_LINK_UNI && !pull) r.member("p2p_gbps", matrix_json(m, w));
    }
    js += "}";
    std::vector<gmd_link_finding_t> f((size_t)w * w * 3 + 2 * w);
    int nf = (int)f.size();
    int rc = gmd_link_diagnose(w, mat[0].data(), mat[1].empty() ? nullptr : mat[1].data(), mat[2].empty() ? nullptr : mat[2].data(),
                               c.verify ? bad_total.data() : nullptr, &fe, f.data(), &nf);
    if (rc < 0) return 2;
    for (int k = 0; k < nf; k++) {
      // the GPU a finding is charged to: the one that ran the copy, or for a slow receiver the peer itself
      const int pos = f[k].src >= 0 ? f[k].src : f[k].dst;
      const int link = f[k].src >= 0 && f[k].dst >= 0 ? map.link_to(f[k].src, f[k].dst) : -1;
      std::string where = gmd_link_finding_name(f[k].kind);
      if (f[k].src >= 0 && f[k].dst >= 0) where += ", link " + (link >= 0 ? std::to_string(link) : std::string("?")) + " to GPU " + std::to_string(ids[f[k].dst]);
      where += std::string(" (") + (pull ? "pull, " : "push, ") + names[f[k].mode] + "): " + f[k].message;
      r.text("  %s GPU %d %s\n", f[k].health >= GM_HEALTH_FAIL ? "FAIL" : "WARN", ids[pos], where.c_str());
      if (f[k].health > worst[pos]) worst[pos] = f[k].health, worst_msg[pos] = where;
      findings += std::string(n_findings++ ? "," : "") + "{\"kind\":\"" + gmd_link_finding_name(f[k].kind) + "\",\"direction\":\"" + (pull ? "pull" : "push") +
                  "\",\"pattern\":\"" + names[f[k].mode] + "\",\"gpu\":" + std::to_string(ids[pos]) + ",\"pci_bdf\":" + jstr(bdf_of_hip(ids[pos])) +
                  ",\"link\":" + (link >= 0 ? std::to_string(link) : std::string("null")) + ",\"peer\":" +
                  (f[k].src >= 0 && f[k].dst >= 0 ? std::to_string(ids[f[k].dst]) : std::string("null")) + ",\"health\":" + std::to_string(f[k].health) +
                  ",\"gbps\":" + jnum(f[k].gbps) + ",\"expected_gbps\":" + jnum(f[k].expected_gbps) + ",\"message\":" + jstr(f[k].message) + "}";
    }
  }
  js += ",\"findings\":" + findings + "]";
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

// ---------------------------------------------------------------- several GPUs at once --
// One host thread per GPU runs the same test at the same time, so the node's power delivery and cooling carry
// all GPUs' load together; afterwards every GPU's rates are held against the median of the others.
using GpuTest = std::function<int(const Args&, Report&, int)>;

int run_on(const Args& a, Report& r, const std::vector<int>& gpus, const GpuTest& fn, bool one_at_a_time = false) {
  if (gpus.size() == 1) {  // one GPU: the report is that GPU's
    r.gpu = gpus[0];
    r.pci_bdf = bdf_of_hip(gpus[0]);
    return fn(a, r, gpus[0]);
  }
  if (r.subs.empty()) {
    r.subs.resize(gpus.size());
    for (size_t i = 0; i < gpus.size(); i++) {
      r.subs[i].gpu = gpus[i];
      r.subs[i].pci_bdf = bdf_of_hip(gpus[i]);  // also: every device touched once from this thread before the others start
      r.subs[i].json = r.json;
      r.subs[i].buffered = true;
    }
  }
  std::vector<int> rcs(gpus.size(), 0);
  std::vector<std::thread> th;
  for (size_t i = 0; i < gpus.size(); i++) {
    th.emplace_back([&, i] { rcs[i] = fn(a, r.subs[i], gpus[i]); });
    if (one_at_a_time) th.back().join();  // tests of something the GPUs share (the host's memory and root ports)
  }
  for (auto& t : th)
    if (t.joinable()) t.join();
  int rc = 0;
  for (size_t i = 0; i < gpus.size(); i++) {
    if (!r.json && !r.subs[i].out.empty()) {
      printf("=== GPU %d (%s)\n%s", gpus[i], r.subs[i].pci_bdf.c_str(), r.subs[i].out.c_str());
      r.subs[i].out.clear();
    }
    if (rcs[i]) rc = rcs[i];
    r.failed = r.failed || r.subs[i].failed;
    r.warned = r.warned || r.subs[i].warned;
  }
  return rc;
}

// Every rate the GPUs measured, each GPU against the median of the others, and what the node drew in total.
void compare_gpus(const Args& a, Report& r) {
  const size_t n = r.subs.size();
  if (n < 2) return;
  const double warn = a.num("--outlier-warn", 0.95), fail = a.num("--outlier-fail", 0.90);
  std::map<std::string, int> keys;
  for (auto& s : r.subs)
    for (auto& kv : s.metrics) keys[kv.first]++;
  std::string cj = "{";
  bool first = true;
  r.text("=== node: every GPU against the median of the others (warn under %.0f %%, fail under %.0f %%)\n", 100 * warn, 100 * fail);
  for (auto& kc : keys) {
    const std::string& key = kc.first;
    if (kc.second < 2 || key == "soak.power_w") continue;
    std::vector<double> v(n, NAN), med(n, 0);
    std::vector<int32_t> h(n, 0);
    for (size_t i = 0; i < n; i++)
      if (r.subs[i].metrics.count(key)) v[i] = r.subs[i].metrics[key];
    gmd_outlier_grade((int)n, v.data(), warn, fail, h.data(), med.data());
    cj += std::string(first ? "" : ",") + "\"" + key + "\":[";
    first = false;
    r.text("  %-22s", key.c_str());
    for (size_t i = 0; i < n; i++) {
      cj += std::string(i ? "," : "") + jnum(v[i]);
      r.text(" %9.4g%s", v[i], h[i] >= GM_HEALTH_FAIL ? "!!" : h[i] ? "! " : "  ");
      if (!std::isfinite(v[i])) continue;
      const uint32_t system = key.rfind("hbm.", 0) == 0 || key.rfind("memtest.", 0) == 0 ? GM_HEALTH_WATCH_MEM
                              : key.rfind("pcie.", 0) == 0                                 ? GM_HEALTH_WATCH_PCIE
                              : key.rfind("soak.", 0) == 0                                 ? GM_HEALTH_WATCH_POWER
                                                                                           : GM_HEALTH_WATCH_COMPUTE;
      if (h[i])
        r.subs[i].verdict("peer_compare", system, h[i], v[i],
                          strprintf_local("%s %.4g is %.1f %% of the other GPUs' median %.4g under the same load", key.c_str(), v[i], 100 * v[i] / med[i], med[i]));
    }
    cj += "]";
    r.text("\n");
  }
  for (auto& s : r.subs) {
    s.verdict("peer_compare", GM_HEALTH_WATCH_COMPUTE, GM_HEALTH_PASS, NAN, "in line with the other GPUs of the node");
    r.failed = r.failed || s.failed;
    r.warned = r.warned || s.warned;
    if (!r.json && !s.out.empty()) fputs(s.out.c_str(), stdout), s.out.clear();
  }
  cj += "}";
  double power = 0, clk_lo = 1e300;
  int np = 0, lo_gpu = -1;
  for (auto& s : r.subs) {
    if (s.metrics.count("soak.power_w")) power += s.metrics["soak.power_w"], np++;
    if (s.metrics.count("soak.gfx_clock_mhz") && s.metrics["soak.gfx_clock_mhz"] < clk_lo) clk_lo = s.metrics["soak.gfx_clock_mhz"], lo_gpu = s.gpu;
  }
  std::string nj = "{\"gpus\":" + std::to_string(n) + ",\"compare\":" + cj;
  if (np) {
    r.text("  node power under load: %.0f W over %d GPUs; lowest sustained clock %.0f MHz on GPU %d\n", power, np, clk_lo, lo_gpu);
    nj += ",\"power_w\":" + jnum(power) + ",\"lowest_clock_mhz\":" + jnum(clk_lo) + ",\"lowest_clock_gpu\":" + std::to_string(lo_gpu);
  }
  r.member("node", nj + "}");
}

// Deployment checks: nothing is loaded, the node is only looked at.  What NVIDIA's diagnostic calls its software level
// (libraries, permissions, device nodes) plus what goes wrong on an 8-GPU AMD node before any kernel runs: a GPU that
// the compute runtime does not see, a device node the user may not open, a host link that trained below what the card
// supports, a lowered power cap, pages waiting for retirement, one GPU on another VBIOS or in another partition mode
// than its seven neighbours, an xGMI link that did not train.  One verdict per GPU and concern; the report is a node
// report.
int cmd_preflight(const Args& a, Report& r) {
  r.gpu = -1;
  const std::string backend = a.str("--backend");
  gm_init_params_t p;
  memset(&p, 0, sizeof p);
  p.version = GM_INIT_PARAMS_VERSION;
  p.mode = GM_MODE_EMBEDDED;
  p.backend = backend.empty() ? nullptr : backend.c_str();
  gm_status_t st = gm_init(&p);
  if (st != GM_ST_OK) {
    r.text("device library: %s\n", gm_strerror(st));
    r.verdict_fail("preflight_runtime", GM_HEALTH_WATCH_DRIVER, 0, "the device library cannot be used: %s", gm_strerror(st));
    return 0;
  }
  uint32_t n = 0;
  gm_device_count(&n);
  std::vector<gm_device_attributes_t> at(n);
  for (uint32_t i = 0; i < n; i++) {
    memset(&at[i], 0, sizeof at[i]);
    at[i].version = GM_DEVICE_ATTRIBUTES_VERSION;
    if (gm_device_attributes(i, &at[i]) != GM_ST_OK) at[i].supported = 0;
  }
  auto verdict = [&](uint32_t g, const char* test, uint32_t system, int health, double value, const std::string& msg) {
    r.verdict_for_gpu((int)g, test, system, health, value, msg);
    for (auto& v : r.verdicts)
      if (v.test == test && v.gpu == (int)g) v.pci_bdf = at[g].pci_bdf;
  };
  auto first_line = [](const std::string& path) {
    std::string s;
    if (FILE* f = fopen(path.c_str(), "r")) {
      char buf[160];
      if (fgets(buf, sizeof buf, f)) s = buf;
      fclose(f);
    }
    while (!s.empty() && (s.back() == '\n' || s.back() == ' ')) s.pop_back();
    return s;
  };
  auto gen_of = [](double gts) { return gts >= 63 ? 6 : gts >= 31 ? 5 : gts >= 15 ? 4 : gts >= 7.5 ? 3 : gts >= 4.5 ? 2 : gts >= 2 ? 1 : 0; };

  // ---- the node: driver, device nodes, the two views of the GPU count
  // the same override the device layer takes: tests lay out a sysfs tree of their own
  const std::string sys = getenv("GPUMON_SYSFS_ROOT") && *getenv("GPUMON_SYSFS_ROOT") ? getenv("GPUMON_SYSFS_ROOT") : "/sys";
  const std::string drv = first_line(sys + "/module/amdgpu/version");
  const bool rocm_host = access("/dev/kfd", F_OK) == 0;
  r.text("device library: %u GPU(s)%s%.48s\n", n, n && at[0].driver_version[0] ? ", driver " : "", n ? at[0].driver_version : "");
  if (!drv.empty()) r.text("amdgpu module: %s\n", drv.c_str());
  std::string node = "{\"gpus\":" + std::to_string(n) + ",\"amdgpu_module\":" + jstr(drv);
  const int want = (int)a.num("--expect-gpus", 0);
  int runtime_health = GM_HEALTH_PASS;
  std::string runtime_msg = "device library and compute runtime agree";
  if (want > 0 && (int)n != want) {
    runtime_health = GM_HEALTH_FAIL;
    runtime_msg = strprintf_local("%u GPU(s) enumerated, %d expected", n, want);
  }
  if (rocm_host && access("/dev/kfd", R_OK | W_OK) != 0) {
    runtime_health = GM_HEALTH_FAIL;
    runtime_msg = strprintf_local("/dev/kfd is not readable and writable by uid %d (render / video group)", (int)getuid());
  }
  if (!a.has("--no-runtime")) {
    const int hip = gmd_device_count();
    node += ",\"runtime_gpus\":" + std::to_string(hip);
    const char* hidden = getenv("HIP_VISIBLE_DEVICES") ? "HIP_VISIBLE_DEVICES" : getenv("ROCR_VISIBLE_DEVICES") ? "ROCR_VISIBLE_DEVICES" : nullptr;
    r.text("compute runtime: %d GPU(s)%s%s\n", std::max(hip, 0), hidden ? ", limited by " : "", hidden ? hidden : "");
    if (hip <= 0) {
      runtime_health = GM_HEALTH_FAIL;
      runtime_msg = strprintf_local("the compute runtime sees no GPU (%s)", gmd_last_error());
    } else if ((uint32_t)hip != n && runtime_health < GM_HEALTH_FAIL) {
      // fewer with a visibility variable set is the operator's doing; anything else is a GPU that one side has lost
      runtime_health = hidden && (uint32_t)hip < n ? GM_HEALTH_WARN : GM_HEALTH_FAIL;
      runtime_msg = strprintf_local("the device library has %u GPU(s), the compute runtime %d%s", n, hip,
                                    hidden ? " (a visibility variable is set)" : " (a device-cgroup rule, or a GPU that fell off the bus)");
    }
  }
  r.verdict("preflight_runtime", GM_HEALTH_WATCH_DRIVER, runtime_health, (double)n, runtime_msg);
  r.member("node", node + "}");

  // ---- memory state of every GPU: one forced sample of the counters that matter before a job starts
  const uint16_t ids[] = {GM_FI_BAD_PAGES_PENDING, GM_FI_BAD_PAGES_UE, GM_FI_ECC_UE_VOLATILE, GM_FI_ECC_CE_VOLATILE};
  std::vector<std::vector<gm_value_t>> vals(n, std::vector<gm_value_t>(4));
  if (n) {
    gm_group_t grp = 0;
    gm_field_group_t fg = 0;
    if (gm_group_create("preflight", &grp) == GM_ST_OK && gm_field_group_create(ids, 4, "preflight", &fg) == GM_ST_OK) {
      for (uint32_t i = 0; i < n; i++) gm_group_add_device(grp, i);
      gm_watch_fields(grp, fg, 1000000, 60.0, 2);
      gm_update_all_fields(1);
      for (uint32_t i = 0; i < n; i++) gm_latest_values(i, ids, 4, vals[i].data());
      gm_unwatch_fields(grp, fg);
      gm_field_group_destroy(fg);
      gm_group_destroy(grp);
    }
  }
  auto num = [](const gm_value_t& v) { return v.status != GM_VS_OK ? -1.0 : v.type == GM_FT_DOUBLE ? v.v.dbl : (double)v.v.i64; };

  // ---- what the GPUs of a node should have in common: the value most of them have is the node's
  struct Common { const char* what; std::function<std::string(const gm_device_attributes_t&)> of; };
  const Common common[] = {{"VBIOS", [](const gm_device_attributes_t& x) { return std::string(x.vbios); }},
                           {"firmware", [](const gm_device_attributes_t& x) { return std::string(x.firmware); }},
                           {"product", [](const gm_device_attributes_t& x) { return std::string(x.name); }},
                           {"compute partition", [](const gm_device_attributes_t& x) { return std::string(x.compute_partition); }},
                           {"memory partition", [](const gm_device_attributes_t& x) { return std::string(x.memory_partition); }},
                           {"HBM size", [](const gm_device_attributes_t& x) { return std::to_string(x.vram_total_mib) + " MiB"; }}};
  std::vector<std::string> odd(n);
  for (const Common& c : common) {
    std::map<std::string, int> count;
    for (uint32_t i = 0; i < n; i++) count[c.of(at[i])]++;
    if (count.size() < 2) continue;
    std::string most;
    for (auto& kv : count)
      if (most.empty() || kv.second > count[most]) most = kv.first;
    for (uint32_t i = 0; i < n; i++)
      if (c.of(at[i]) != most) odd[i] += (odd[i].empty() ? "" : "; ") + std::string(c.what) + " " + c.of(at[i]) + ", most GPUs of the node have " + most;
  }

  r.text("%-3s %-13s %-22s %-14s %-9s %-5s %-14s %-11s %s\n", "gpu", "pci", "product", "vbios", "partition", "numa", "host link", "power cap", "xgmi  pages");
  std::string arr = "[";
  for (uint32_t i = 0; i < n; i++) {
    const gm_device_attributes_t& x = at[i];
    // device node
    int access_health = GM_HEALTH_PASS;
    std::string access_msg = "device node can be opened";
    if (!x.supported) access_health = GM_HEALTH_WARN, access_msg = "not a device this stack samples in full (not a CDNA-class GPU)";
    if (x.device_path[0] && (rocm_host || access(x.device_path, F_OK) == 0) && access(x.device_path, R_OK | W_OK) != 0) {
      access_health = GM_HEALTH_FAIL;
      access_msg = strprintf_local("%s is not readable and writable by uid %d (render / video group)", x.device_path, (int)getuid());
    }
    verdict(i, "preflight_access", GM_HEALTH_WATCH_DRIVER, access_health, 0, access_msg);
    // host link: what it trained to against what the card supports
    const std::string dir = sys + "/bus/pci/devices/" + x.pci_bdf + "/";
    const int cur_gen = gen_of(atof(first_line(dir + "current_link_speed").c_str())), cur_w = atoi(first_line(dir + "current_link_width").c_str());
    int max_gen = gen_of(atof(first_line(dir + "max_link_speed").c_str())), max_w = atoi(first_line(dir + "max_link_width").c_str());
    if (max_gen <= 0) max_gen = (int)x.pcie_max_gen;
    if (max_w <= 0) max_w = (int)x.pcie_max_width;
    std::string link = cur_gen > 0 && cur_w > 0 ? strprintf_local("gen%d x%d", cur_gen, cur_w) : "-";
    if (cur_gen > 0 && cur_w > 0) {
      const bool low = (max_gen > 0 && cur_gen < max_gen) || (max_w > 0 && cur_w < max_w);
      if (low) link += strprintf_local(" of gen%d x%d", max_gen, max_w);
      verdict(i, "preflight_pcie", GM_HEALTH_WATCH_PCIE, low ? GM_HEALTH_WARN : GM_HEALTH_PASS, cur_gen * 100 + cur_w,
              low ? strprintf_local("host link trained to generation %d x%d, the device supports generation %d x%d (slot, riser or reseat)", cur_gen, cur_w, max_gen, max_w)
                  : "host link at the generation and width the device supports");
    }
    // power cap
    std::string cap = std::isfinite(x.power_cap_w) ? strprintf_local("%.0f W", x.power_cap_w) : "-";
    if (std::isfinite(x.power_cap_w) && std::isfinite(x.power_cap_default_w) && x.power_cap_default_w > 0) {
      const bool low = x.power_cap_w < x.power_cap_default_w - 0.5;
      if (low) cap += strprintf_local(" of %.0f", x.power_cap_default_w);
      verdict(i, "preflight_power", GM_HEALTH_WATCH_POWER, low ? GM_HEALTH_WARN : GM_HEALTH_PASS, x.power_cap_w,
              low ? strprintf_local("power cap %.0f W is under the default %.0f W", x.power_cap_w, x.power_cap_default_w) : "power cap at its default");
    }
    // memory
    const double pending = num(vals[i][0]), retired_ue = num(vals[i][1]), ue = num(vals[i][2]), ce = num(vals[i][3]);
    std::string pages = pending < 0 && retired_ue < 0 ? "-" : strprintf_local("%.0f pending, %.0f retired", std::max(pending, 0.0), std::max(retired_ue, 0.0));
    if (pending >= 0 || ue >= 0) {
      int h = GM_HEALTH_PASS;
      std::string msg = "no uncorrectable errors since the driver was loaded, no pages waiting for retirement";
      if (pending > 0) h = GM_HEALTH_WARN, msg = strprintf_local("%.0f page(s) wait for retirement at the next driver load", pending);
      if (ue > 0) h = GM_HEALTH_FAIL, msg = strprintf_local("%.0f uncorrectable ECC error(s) since the driver was loaded", ue);
      verdict(i, "preflight_memory", GM_HEALTH_WATCH_MEM, h, std::max(ue, 0.0) + std::max(pending, 0.0), msg);
    }
    // fabric: every wired xGMI link is up (a link that did not train takes its traffic over the others, or over the host)
    int links_up = -1, links_wired = 0;
    if (n > 1) {
      gm_xgmi_link_t lk[GM_MAX_XGMI_LINKS];
      int nl = GM_MAX_XGMI_LINKS;
      if (gm_xgmi_links(i, lk, &nl) == GM_ST_OK && nl > 0) {
        links_up = 0;
        std::string down;
        for (int l = 0; l < nl; l++) {
          if (lk[l].state < 0) continue;
          links_wired++;
          if (lk[l].state == 1) { links_up++; continue; }
          down += (down.empty() ? "" : ", ") + std::to_string(lk[l].link);
          if (lk[l].peer_gpu >= 0) down += " (to GPU " + std::to_string(lk[l].peer_gpu) + ")";
          else if (lk[l].peer_bdf[0]) down += std::string(" (to ") + lk[l].peer_bdf + ")";
        }
        if (links_wired)
          verdict(i, "preflight_fabric", GM_HEALTH_WATCH_XGMI, down.empty() ? GM_HEALTH_PASS : GM_HEALTH_FAIL, links_up,
                  down.empty() ? strprintf_local("%d of %d xGMI links up", links_up, links_wired)
                               : strprintf_local("xGMI link %s down: %d of %d links up", down.c_str(), links_up, links_wired));
      }
    }
    // the odd one out
    if (n > 1) verdict(i, "preflight_uniform", GM_HEALTH_WATCH_FIRMWARE, odd[i].empty() ? GM_HEALTH_PASS : GM_HEALTH_WARN, 0,
                       odd[i].empty() ? "VBIOS, firmware, product, partition modes and HBM size as on the other GPUs" : odd[i]);
    const std::string part = std::string(x.compute_partition[0] ? x.compute_partition : "-") + "/" + (x.memory_partition[0] ? x.memory_partition : "-");
    const std::string fabric = links_up < 0 || !links_wired ? "-" : strprintf_local("%d/%d", links_up, links_wired);
    r.text("%-3u %-13s %-22.22s %-14.14s %-9s %-5d %-14s %-11s %-5s %s\n", i, x.pci_bdf, x.name, x.vbios[0] ? x.vbios : "-", part.c_str(), x.numa_node, link.c_str(),
           cap.c_str(), fabric.c_str(), pages.c_str());
    arr += std::string(i ? "," : "") + "{\"gpu\":" + std::to_string(i) + ",\"pci_bdf\":" + jstr(x.pci_bdf) + ",\"name\":" + jstr(x.name) + ",\"vbios\":" + jstr(x.vbios) +
           ",\"firmware\":" + jstr(x.firmware) + ",\"compute_partition\":" + jstr(x.compute_partition) + ",\"memory_partition\":" + jstr(x.memory_partition) +
           ",\"numa_node\":" + std::to_string(x.numa_node) + ",\"link_gen\":" + std::to_string(cur_gen) + ",\"link_width\":" + std::to_string(cur_w) +
           ",\"link_max_gen\":" + std::to_string(max_gen) + ",\"link_max_width\":" + std::to_string(max_w) + ",\"power_cap_w\":" + jnum(x.power_cap_w) +
           ",\"power_cap_default_w\":" + jnum(x.power_cap_default_w) + ",\"xgmi_links_up\":" + (links_up < 0 ? std::string("null") : std::to_string(links_up)) +
           ",\"xgmi_links_wired\":" + std::to_string(links_wired) + ",\"pages_pending\":" + jnum(pending < 0 ? NAN : pending) +
           ",\"ecc_ue\":" + jnum(ue < 0 ? NAN : ue) + ",\"ecc_ce\":" + jnum(ce < 0 ? NAN : ce) + "}";
  }
  r.member("devices", arr + "]");
  gm_shutdown();
  return 0;
}


// gpumon-diag expect [--arch gfx950] [--name "AMD Instinct MI355X"] [--expect ..] [--expect-scale F] [--check key=value,..]
// What a GPU is held against, without needing one: the table entry for an architecture / product name (default:
// GPU 0 when one is visible, else gfx950), and with --check the grade of given measurements.
int cmd_expect(const Args& a, Report& r) {
  std::string arch = a.str("--arch"), name = a.str("--name");
  if (arch.empty()) {
    gmd_device_info_t d;
    if (gmd_device_count() > 0 && gmd_device_info((int)a.num("-g", 0), &d) == GMD_OK) arch = d.gcn_arch, name = d.name;
    else arch = "gfx950";
  }
  diag_expect::Expect e = a.has("--no-expect") ? diag_expect::Expect() : diag_expect::lookup(arch, name);
  std::string bad;
  if (a.has("--expect") && !diag_expect::apply_overrides(a.str("--expect"), &e, &bad)) {
    fprintf(stderr, "--expect: cannot read '%s' (key=value, value >= 0)\n", bad.c_str());
    return 2;
  }
  e.scale = a.num("--expect-scale", 1.0);
  r.text("expectations for %s %s: %s (warn under %.0f %%, fail under %.0f %%, scale %.2f)\n", arch.c_str(), name.c_str(),
         e.known ? e.label.c_str() : "none: correctness is graded, performance only reported", 100 * e.warn_ratio, 100 * e.fail_ratio, e.scale);
  std::string tj = "{";
  bool first = true;
  for (auto& kv : e.v) {
    const double v = e.want(kv.first);
    r.text("  %-32s %10.4g\n", kv.first.c_str(), v);
    tj += std::string(first ? "" : ",") + "\"" + kv.first + "\":" + jnum(v);
    first = false;
  }
  r.member("expect", "{\"arch\":" + jstr(arch) + ",\"known\":" + (e.known ? "true" : "false") + ",\"warn_ratio\":" + jnum(e.warn_ratio) +
                         ",\"fail_ratio\":" + jnum(e.fail_ratio) + ",\"scale\":" + jnum(e.scale) + ",\"table\":" + tj + "}}");
  if (a.has("--check")) {
    diag_expect::Expect m;
    if (!diag_expect::apply_overrides(a.str("--check"), &m, &bad)) {
      fprintf(stderr, "--check: cannot read '%s'\n", bad.c_str());
      return 2;
    }
    std::string cj = "{";
    first = true;
    for (auto& kv : m.v) {
      const uint32_t system = kv.first.rfind("hbm.", 0) == 0 || kv.first.rfind("memtest.", 0) == 0 ? GM_HEALTH_WATCH_MEM
                              : kv.first.rfind("pcie.", 0) == 0                                    ? GM_HEALTH_WATCH_PCIE
                              : kv.first.rfind("soak.", 0) == 0                                    ? GM_HEALTH_WATCH_POWER
                                                                                                    : GM_HEALTH_WATCH_COMPUTE;
      const std::string test = kv.first.substr(0, kv.first.find('.'));
      const int h = r.grade(test, system, kv.first, kv.second, e.want(kv.first), e, "");
      r.verdict_pass(test, system, kv.second, "at its expected rate");
      r.text("  check %-26s %10.4g -> %s\n", kv.first.c_str(), kv.second, diag_expect::health_name(h));
      cj += std::string(first ? "" : ",") + "\"" + kv.first + "\":" + std::to_string(h);
      first = false;
    }
    r.member("check", cj + "}");
  }
  return 0;
}

int cmd_run(const Args& a, Report& r, const std::vector<int>& gpus, MfmaWatch* mfma) {
  const int level = (int)a.num("-r", 1);
  if (level < 1 || level > 3) {
    fprintf(stderr, "-r takes 1, 2 or 3\n");
    return 2;
  }
  Args sub = a;
  int rc;
  if ((rc = cmd_info(sub, r))) return rc;
  // level 1 (quick, seconds): short stream + 1 GiB memtest.  level 2 adds the matrix pipes and a 5 s
  // GEMM stress.  level 3 sweeps 90 % of HBM, soaks 60 s and exercises the fabric (once, over the GPUs of the run).
  // With several GPUs every step runs on all of them at the same time.
  sub.kv["--bytes"] = level == 1 ? "1G" : "4G";
  sub.kv["--iters"] = level == 1 ? "5" : "20";
  if ((rc = run_on(sub, r, gpus, [](const Args& x, Report& y, int g) { return cmd_hbm(x, y, g); }))) return rc;
  {
    // the host link: one GPU after the other, since the GPUs of a node share the host's memory and root ports
    Args host = sub;
    host.kv["--bytes"] = level == 1 ? "64M" : "256M";
    host.kv["--iters"] = level == 1 ? "3" : "5";
    if ((rc = run_on(host, r, gpus, [](const Args& x, Report& y, int g) { return cmd_pcie(x, y, g); }, /*one_at_a_time=*/true))) return rc;
  }
  if (level < 3) sub.kv["--bytes"] = level == 1 ? "1G" : "16G";
  else sub.kv.erase("--bytes");
  if (level >= 2) sub.kv["--march"] = "1";
  if ((rc = run_on(sub, r, gpus, [](const Args& x, Report& y, int g) { return cmd_memtest(x, y, g); }))) return rc;
  if (level >= 2) {
    sub.kv["--seconds"] = "1";
    if ((rc = run_on(sub, r, gpus, [](const Args& x, Report& y, int g) { return cmd_mfma(x, y, g); }))) return rc;
    sub.kv["--seconds"] = a.has("--seconds") ? a.str("--seconds") : level == 2 ? "5" : "60";
    sub.kv.erase("--dtype");
    if ((rc = run_on(sub, r, gpus, [mfma](const Args& x, Report& y, int g) { return cmd_gemm(x, y, g, mfma); }))) return rc;
  }
  compare_gpus(a, r);
  if (level >= 3 && !a.has("--no-fabric")) {
    sub.kv.erase("--bytes");
    sub.kv.erase("--iters");
    sub.kv.erase("--seconds");
    sub.kv["--scaling"] = "1";
    // the fabric tests are about the GPUs of this run: all of them for a node run, the visible ones for one GPU
    if (gpus.size() > 1) {
      std::string list;
      for (size_t i = 0; i < gpus.size(); i++) list += (i ? "," : "") + std::to_string(gpus[i]);
      sub.kv["--gpus"] = list;
    } else sub.kv.erase("--gpus");
    if ((rc = cmd_rccl(sub, r))) return rc;
    if ((gpus.size() > 1 || gmd_device_count() > 1) && (rc = cmd_links(sub, r))) return rc;
  }
  return 0;
}

void usage() {
  fprintf(stderr,
          "usage: gpumon-diag <info|preflight|expect|hbm|memtest|pcie|mfma|gemm|rccl|links|run> [options] [--json] [--publish FILE]\n"
          "  --publish FILE leaves the JSON report where `gpumon-exporter --diag-results` picks it up\n"
          "  preflight [--expect-gpus N] [--no-runtime] [--backend amdsmi|fake]   deployment checks; loads nothing\n"
          "  hbm     [-g GPU | --gpus LIST|all] [--bytes N] [--iters N] [--kernel copy|scale|add|triad|read|write|all] [--min-tbps X]\n"
          "  memtest [-g GPU | --gpus LIST|all] [--bytes N | --fraction F] [--chunk N] [--patterns LIST] [--passes N] [--march]\n"
          "  pcie    [-g GPU | --gpus LIST|all] [--bytes N] [--iters N] [--pcie-gen G --pcie-width W]   host link: copy engines and kernels over pinned memory\n"
          "  mfma    [-g GPU | --gpus LIST|all] [--dtype bf16|fp8|mxfp8|f32|all] [--seconds S]\n"
          "  gemm    [-g GPU | --gpus LIST|all] [--dtype LIST] [--m M --n N --k K] [--seconds S] [--check-every N] [--min-tflops X]\n"
          "          [--no-telemetry] [--mfma-counters]\n"
          "  rccl    [--gpus LIST] [--ops LIST] [--min-bytes N] [--max-bytes N] [--factor N] [--iters N] [--scaling] [--no-verify] [--min-busbw GBPS]\n"
          "  links   [--gpus LIST] [--bytes N] [--iters N] [--modes uni,bidi,all] [--direction push|pull|both] [--link-gbps X] [--floor-gbps X]\n"
          "          [--warn-ratio R] [--fail-ratio R]\n"
          "          peer-copy bandwidth of every xGMI link (remote stores and remote loads), read into findings per GPU, link and peer\n"
          "  run     -r 1|2|3 [-g GPU | --gpus LIST|all] [--no-fabric]\n"
          "  expect  [--arch ARCH] [--name PRODUCT] [--check KEY=VALUE,..]   the table a GPU is held against; grades given measurements\n"
          "grading: throughputs against the expectation table of the GPU model (PASS, WARN under 90 %%, FAIL under 75 %%);\n"
          "         [--expect key=value,..] [--expect-scale F] [--no-expect]; with --gpus also each GPU against the median of the\n"
          "         others [--outlier-warn 0.95] [--outlier-fail 0.90]\n"
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
  // HIP device index -> the engine's index of the same card; -1 when the engine does not have it
  auto engine_index = [](int hip) {
    uint32_t n = 0;
    const std::string bdf = bdf_of_hip(hip);
    if (!bdf.empty() && gm_device_count(&n) == GM_ST_OK)
      for (uint32_t i = 0; i < n; i++) {
        gm_device_attributes_t at;
        memset(&at, 0, sizeof at);
        at.version = GM_DEVICE_ATTRIBUTES_VERSION;
        if (gm_device_attributes(i, &at) == GM_ST_OK && strcasecmp(at.pci_bdf, bdf.c_str()) == 0) return (int)i;
      }
    return -1;
  };
  bool ok = true;
  const Report all = r.all_verdicts();
  for (auto& v : all.verdicts) {
    const int hip = v.gpu >= 0 ? v.gpu : r.gpu;
    const int idx = hip >= 0 ? engine_index(hip) : -1;
    if (idx < 0) {
      fprintf(stderr, "warning: the %s verdict is about HIP device %d (%s), which the host engine does not list; not submitted\n", v.test.c_str(), hip,
              hip >= 0 ? bdf_of_hip(hip).c_str() : "node");
      continue;
    }
    gm_diag_result_t dr;
    memset(&dr, 0, sizeof dr);
    dr.version = GM_DIAG_RESULT_VERSION;
    dr.gpu = (uint32_t)idx;
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
  // The matrix-pipe counter probe is a child process and must exist before this process touches the device.
  MfmaWatch mfma;
  const bool wants_mfma = a.has("--mfma-counters") && (cmd == "gemm" || cmd == "run");
  const bool have_mfma = wants_mfma && mfma.start();
  if (wants_mfma && !have_mfma) fprintf(stderr, "note: gpumon-prof-probe not found next to this binary; no matrix-pipe counters\n");
  MfmaWatch* mw = have_mfma ? &mfma : nullptr;
  int rc;
  const bool per_gpu = cmd == "hbm" || cmd == "memtest" || cmd == "pcie" || cmd == "mfma" || cmd == "gemm" || cmd == "run";
  std::vector<int> gpus;
  if (per_gpu) {
    gpus = target_gpus(a);
    if (gpus.empty()) {
      fprintf(stderr, "error: no GPU to test: %s\n", gmd_last_error());
      return 2;
    }
  }
  if (cmd == "info") rc = cmd_info(a, r);
  else if (cmd == "preflight") rc = cmd_preflight(a, r);
  else if (cmd == "expect") rc = cmd_expect(a, r);
  else if (cmd == "hbm") rc = run_on(a, r, gpus, [](const Args& x, Report& y, int g) { return cmd_hbm(x, y, g); });
  else if (cmd == "memtest") rc = run_on(a, r, gpus, [](const Args& x, Report& y, int g) { return cmd_memtest(x, y, g); });
  else if (cmd == "pcie") rc = run_on(a, r, gpus, [](const Args& x, Report& y, int g) { return cmd_pcie(x, y, g); }, /*one_at_a_time=*/true);
  else if (cmd == "mfma") rc = run_on(a, r, gpus, [](const Args& x, Report& y, int g) { return cmd_mfma(x, y, g); });
  else if (cmd == "gemm") rc = run_on(a, r, gpus, [mw](const Args& x, Report& y, int g) { return cmd_gemm(x, y, g, mw); });
  else if (cmd == "rccl") rc = cmd_rccl(a, r);
  else if (cmd == "links" || cmd == "p2p") rc = cmd_links(a, r);
  else if (cmd == "run") rc = cmd_run(a, r, gpus, mw);
  else {
    usage();
    return 2;
  }
  if (rc) return rc;
  if (per_gpu && cmd != "run") compare_gpus(a, r);
  if (have_mfma) {
    mfma.stop();
    for (int g : gpus) r.member(gpus.size() > 1 ? "mfma_active_gpu" + std::to_string(g) : "mfma_active", mfma.fold(r, bdf_of_hip(g).c_str()));
  }
  rc = r.finish(a.str("--publish"), bdf_of_hip);
  if (rc != 2 && a.has("--submit") && !submit_verdicts(a, r)) return 2;
  return rc;
}
