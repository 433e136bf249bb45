// Active-diagnostics results as Prometheus metrics.
//
// `gpumon-diag ... --publish FILE` leaves its JSON report behind; the exporter
// (--diag-results FILE|DIR) turns every report it finds into gpumon_diag_*
// families, so a failed memtest, a GEMM that lost its checksum, an HBM stack
// that streams below its floor or an xGMI pair far under its siblings becomes
// an ordinary alerting-rule input next to the passive telemetry.  Reports are
// re-parsed only when their mtime changes.
#include <dirent.h>
#include <sys/stat.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>

#include "../common/json.h"
#include "../common/util.h"
#include "exporter.h"

namespace gpumon {

namespace {

// family name -> formatted label set (gpu="0",kernel="copy") -> value.  Keyed by
// label set so that node-wide series (RCCL, peer copies) found in the reports of
// several GPUs come out once: the report read last wins.
using Families = std::map<std::string, std::map<std::string, double>>;

struct CachedReport {
  int64_t mtime_ns = -1;
  Families fams;
};

const char* help_of(const std::string& fam) {
  static const std::map<std::string, const char*> kHelp = {
      {"gpumon_diag_pass", "1 if the last active-diagnostics run on this GPU passed, 0 if any test failed."},
      {"gpumon_diag_timestamp_seconds", "Unix time of the last active-diagnostics report."},
      {"gpumon_diag_hbm_stream_tbps", "HBM stream bandwidth measured by the diagnostics (in TB/s)."},
      {"gpumon_diag_memtest_errors", "Mismatching 64-bit words found by the HBM pattern test."},
      {"gpumon_diag_memtest_bytes", "Bytes of HBM covered by the pattern test."},
      {"gpumon_diag_mfma_peak_tflops", "Matrix-pipe peak measured on register operands (in TFLOP/s)."},
      {"gpumon_diag_gemm_tflops", "GEMM stress throughput on random data (in TFLOP/s)."},
      {"gpumon_diag_gemm_sdc_mismatches", "GEMM launches whose output checksum changed (silent data corruption)."},
      {"gpumon_diag_soak_power_watts", "Board power during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_soak_gfx_clock_mhz", "Shader clock held during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_soak_gpu_temp_celsius", "Hotspot temperature during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_gemm_mfma_active_percent", "Matrix (MFMA) pipe busy during the GEMM stress of one data type, from the hardware counter (average)."},
      {"gpumon_diag_soak_mfma_active_percent", "Matrix (MFMA) pipe busy during the GEMM soak, from the hardware counter (min / avg / max)."},
      {"gpumon_diag_rccl_busbw_gbps", "Peak RCCL bus bandwidth per collective (in GB/s)."},
      {"gpumon_diag_p2p_gbps", "Peer copy bandwidth between two GPUs (in GB/s)."},
  };
  auto it = kHelp.find(fam);
  return it == kHelp.end() ? "Active diagnostics result." : it->second;
}

double num(const Json* j, double dflt = 0) {
  if (!j) return dflt;
  if (j->type == Json::NUM) return j->num;
  if (j->type == Json::BOOL) return j->b ? 1 : 0;
  return dflt;
}

void parse_report(const Json& r, Families* f) {
  const std::string gpu = strprintf("gpu=\"%d\"", (int)num(r.get("gpu"), 0));
  auto add = [&](const char* fam, const std::string& extra, double v) {
    (*f)[fam][extra.empty() ? gpu : gpu + "," + extra] = v;
  };
  if (const Json* res = r.get("result")) add("gpumon_diag_pass", "", res->type == Json::STR && res->str == "pass" ? 1 : 0);
  if (const Json* ts = r.get("timestamp")) add("gpumon_diag_timestamp_seconds", "", num(ts));
  if (const Json* hbm = r.get("hbm_stream"))
    for (auto& kv : hbm->obj) add("gpumon_diag_hbm_stream_tbps", "kernel=\"" + kv.first + "\"", num(kv.second.get("tbps_best")));
  if (const Json* m = r.get("memtest")) {
    add("gpumon_diag_memtest_errors", "", num(m->get("errors")));
    add("gpumon_diag_memtest_bytes", "", num(m->get("bytes_tested")));
  }
  if (const Json* p = r.get("mfma_peak"))
    for (auto& kv : p->obj) add("gpumon_diag_mfma_peak_tflops", "dtype=\"" + kv.first + "\"", num(kv.second.get("tflops")));
  if (const Json* g = r.get("gemm_stress"))
    for (auto& kv : g->obj) {
      add("gpumon_diag_gemm_tflops", "dtype=\"" + kv.first + "\"", num(kv.second.get("tflops")));
      add("gpumon_diag_gemm_sdc_mismatches", "dtype=\"" + kv.first + "\"", num(kv.second.get("sdc_mismatches")));
      if (kv.second.get("mfma_active_avg"))
        add("gpumon_diag_gemm_mfma_active_percent", "dtype=\"" + kv.first + "\"", num(kv.second.get("mfma_active_avg")));
    }
  if (const Json* t = r.get("telemetry"))
    if (t->type == Json::OBJ) {
      const struct { const char* key; const char* fam; } kSoak[] = {{"power", "gpumon_diag_soak_power_watts"},
                                                                   {"gfx_clock", "gpumon_diag_soak_gfx_clock_mhz"},
                                                                   {"gpu_temp", "gpumon_diag_soak_gpu_temp_celsius"}};
      for (auto& s : kSoak)
        if (const Json* v = t->get(s.key))
          for (const char* stat : {"min", "avg", "max"})
            if (v->get(stat)) add(s.fam, std::string("stat=\"") + stat + "\"", num(v->get(stat)));
    }
  if (const Json* v = r.get("mfma_active"))
    if (v->type == Json::OBJ)
      for (const char* stat : {"min", "avg", "max"})
        if (v->get(stat)) add("gpumon_diag_soak_mfma_active_percent", std::string("stat=\"") + stat + "\"", num(v->get(stat)));
  auto rccl = [&](const Json& sweep) {
    const int n = (int)num(sweep.get("gpus"), 0);
    if (const Json* pk = sweep.get("peak_busbw_gbps"))
      for (auto& kv : pk->obj)
        (*f)["gpumon_diag_rccl_busbw_gbps"][strprintf("gpus=\"%d\",op=\"%s\"", n, kv.first.c_str())] = num(&kv.second);
  };
  if (const Json* s = r.get("rccl")) rccl(*s);
  if (const Json* s = r.get("rccl_scaling"))
    for (auto& e : s->arr) rccl(e);
  if (const Json* m = r.get("p2p_gbps"))
    for (size_t i = 0; i < m->arr.size(); i++)
      for (size_t j = 0; j < m->arr[i].arr.size(); j++)
        if (i != j) (*f)["gpumon_diag_p2p_gbps"][strprintf("src=\"%zu\",dst=\"%zu\"", i, j)] = num(&m->arr[i].arr[j]);
}

std::mutex g_mu;
std::map<std::string, CachedReport> g_cache;

// The per-test verdicts of a report, as the engine takes them.
void submit_verdicts(const Json& r, Engine* engine) {
  const Json* vs = r.get("verdicts");
  if (!engine || !vs || vs->type != Json::ARR) return;
  for (auto& v : vs->arr) {
    if (v.type != Json::OBJ) continue;
    gm_diag_result_t d;
    memset(&d, 0, sizeof d);
    d.version = GM_DIAG_RESULT_VERSION;
    // a multi-GPU test (the link test) names the GPU in each verdict; the others are about the report's GPU
    d.gpu = (uint32_t)num(v.get("gpu") ? v.get("gpu") : r.get("gpu"), 0);
    const Json* t = v.get("test");
    const Json* m = v.get("message");
    if (!t || t->type != Json::STR || t->str.empty()) continue;
    copy_str(d.test, t->str);
    if (m && m->type == Json::STR) copy_str(d.message, m->str);
    d.system = (uint32_t)num(v.get("system"), 0);
    d.health = (int32_t)num(v.get("health"), 0);
    const Json* val = v.get("value");
    d.value = val && val->type == Json::NUM ? val->num : std::nan("");
    d.ts_us = (int64_t)(num(r.get("timestamp"), 0) * 1e6);  // the same report read again is the same verdict
    gm_status_t st = engine->diag_submit(d);
    if (st != GM_ST_OK) GM_LOGW("exporter: diagnostics verdict %s of GPU %u not accepted (%d)", d.test, d.gpu, (int)st);
  }
}

void load(const std::string& file, Families* merged, Engine* engine) {
  struct stat st;
  if (stat(file.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return;
  const int64_t mt = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  CachedReport& c = g_cache[file];
  if (c.mtime_ns != mt) {
    c.mtime_ns = mt;
    c.fams.clear();
    std::string text;
    Json j;
    if (read_file(file, &text) && json_parse(text, &j) && j.type == Json::OBJ) {
      parse_report(j, &c.fams);
      submit_verdicts(j, engine);
    } else GM_LOGW("exporter: %s is not a diagnostics report", file.c_str());
  }
  for (auto& kv : c.fams) {
    auto& dst = (*merged)[kv.first];
    for (auto& s : kv.second) dst[s.first] = s.second;
  }
}

}  // namespace

void render_diag_results(const std::string& path, std::string* out, Engine* engine) {
  if (path.empty()) return;
  std::lock_guard<std::mutex> l(g_mu);
  Families merged;
  struct stat st;
  if (stat(path.c_str(), &st) != 0) return;  // no report yet: nothing to say
  if (S_ISDIR(st.st_mode)) {
    std::vector<std::string> files;
    if (DIR* d = opendir(path.c_str())) {
      while (dirent* e = readdir(d)) {
        std::string n = e->d_name;
        if (n.size() > 5 && n.compare(n.size() - 5, 5, ".json") == 0) files.push_back(path + "/" + n);
      }
      closedir(d);
    }
    std::sort(files.begin(), files.end());
    for (auto& f : files) load(f, &merged, engine);
  } else {
    load(path, &merged, engine);
  }
  for (auto& kv : merged) {
    *out += strprintf("# HELP %s %s\n# TYPE %s gauge\n", kv.first.c_str(), help_of(kv.first), kv.first.c_str());
    // %.15g keeps Unix timestamps and byte counts exact.
    for (auto& s : kv.second) *out += strprintf("%s{%s} %.15g\n", kv.first.c_str(), s.first.c_str(), s.second);
  }
}

}  // namespace gpumon
