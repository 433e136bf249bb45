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

struct Pending {  // one verdict waiting for the engine, with the card's identity as the report gave it
  gm_diag_result_t d;
  std::string pci_bdf;
  int report_gpu = -1;
};

struct CachedReport {
  int64_t mtime_ns = -1;
  Families fams;
  std::vector<Pending> verdicts;
  bool submitted_ok = false;      // every verdict was accepted by the engine
  int64_t submitted_mono_us = 0;  // when that last happened
  int64_t attempt_mono_us = 0;    // when they were last offered, accepted or not
  bool said = false;              // what there is to complain about in this report has been said once
};

const char* help_of(const std::string& fam) {
  static const std::map<std::string, const char*> kHelp = {
      {"gpumon_diag_pass", "1 if the last active-diagnostics run on this GPU passed, 0 if any test failed."},
      {"gpumon_diag_timestamp_seconds", "Unix time of the last active-diagnostics report."},
      {"gpumon_diag_node_pass", "1 if the last node-wide diagnostics report of this name (fabric tests, all GPUs at once) passed, 0 if any test failed."},
      {"gpumon_diag_node_timestamp_seconds", "Unix time of the last node-wide diagnostics report of this name."},
      {"gpumon_diag_test_health", "Verdict of one diagnostic test on one GPU (0 = pass, 10 = warning, 20 = failure)."},
      {"gpumon_diag_hbm_stream_tbps", "HBM stream bandwidth measured by the diagnostics (in TB/s)."},
      {"gpumon_diag_pcie_gbps", "Host link bandwidth measured by the diagnostics over pinned memory, by traffic pattern (in GB/s)."},
      {"gpumon_diag_pcie_link_gbps", "What one direction of the negotiated PCIe link carries at most (generation x width, in GB/s)."},
      {"gpumon_diag_memtest_errors", "Mismatching 64-bit words found by the HBM pattern test."},
      {"gpumon_diag_memtest_bytes", "Bytes of HBM covered by the pattern test."},
      {"gpumon_diag_mfma_peak_tflops", "Matrix-pipe peak measured on register operands (in TFLOP/s)."},
      {"gpumon_diag_gemm_tflops", "GEMM stress throughput on random data (in TFLOP/s)."},
      {"gpumon_diag_gemm_sdc_mismatches", "GEMM launches whose output checksum changed (silent data corruption)."},
      {"gpumon_diag_gemm_checked_launches", "GEMM launches of the stress whose output was held against the first launch's: summed in the kernel, or re-read from memory."},
      {"gpumon_diag_soak_power_watts", "Board power during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_soak_gfx_clock_mhz", "Shader clock held during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_soak_gpu_temp_celsius", "Hotspot temperature during the GEMM soak (min / avg / max)."},
      {"gpumon_diag_gemm_mfma_active_percent", "Matrix (MFMA) pipe busy during the GEMM stress of one data type, from the hardware counter (average)."},
      {"gpumon_diag_soak_mfma_active_percent", "Matrix (MFMA) pipe busy during the GEMM soak, from the hardware counter (min / avg / max)."},
      {"gpumon_diag_rccl_busbw_gbps", "Peak RCCL bus bandwidth per collective (in GB/s)."},
      {"gpumon_diag_rccl_expected_busbw_gbps", "Bus bandwidth expected of the collective on this many GPUs, from the link rate (in GB/s)."},
      {"gpumon_diag_p2p_gbps", "Peer copy bandwidth between two GPUs (in GB/s)."},
      {"gpumon_diag_link_findings", "Findings of the xGMI link test charged to this GPU, by kind (slow_link, half_duplex, low_egress, ...)."},
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
std::string str(const Json* j) { return j && j->type == Json::STR ? j->str : ""; }

// The engine's index of the card a report (or a verdict) is about.  Reports are written by a process that numbers
// GPUs the way the compute runtime does; the engine follows the PCI bus, and a diagnostics container may see a
// subset of the cards.  The PCI address decides where there is one; -1 = the engine has no such card.
struct Resolver {
  Engine* engine;
  std::map<std::string, int> by_bdf;
  bool loaded = false;
  int operator()(const std::string& bdf, int report_gpu) {
    if (!engine || bdf.empty()) return report_gpu;
    if (!loaded) {
      loaded = true;
      uint32_t n = 0;
      if (engine->device_count(&n) == GM_ST_OK)
        for (uint32_t i = 0; i < n; i++) {
          gm_device_attributes_t a;
          memset(&a, 0, sizeof a);
          a.version = GM_DEVICE_ATTRIBUTES_VERSION;
          if (engine->device_attributes(i, &a) != GM_ST_OK) continue;
          std::string b = a.pci_bdf;
          for (auto& c : b) c = (char)tolower((unsigned char)c);
          by_bdf[b] = (int)i;
        }
    }
    std::string b = bdf;
    for (auto& c : b) c = (char)tolower((unsigned char)c);
    auto it = by_bdf.find(b);
    return it == by_bdf.end() ? -1 : it->second;
  }
};

// One per-GPU document: everything measured on that GPU, labelled with the engine's index of the card.
void parse_gpu_report(const Json& r, int gpu_index, Families* f) {
  const std::string gpu = strprintf("gpu=\"%d\"", gpu_index);
  auto add = [&](const char* fam, const std::string& extra, double v) {
    (*f)[fam][extra.empty() ? gpu : gpu + "," + extra] = v;
  };
  if (const Json* res = r.get("result")) add("gpumon_diag_pass", "", res->type == Json::STR && res->str != "fail" ? 1 : 0);
  if (const Json* ts = r.get("timestamp")) add("gpumon_diag_timestamp_seconds", "", num(ts));
  if (const Json* hbm = r.get("hbm_stream"))
    for (auto& kv : hbm->obj) add("gpumon_diag_hbm_stream_tbps", "kernel=\"" + kv.first + "\"", num(kv.second.get("tbps_best")));
  if (const Json* pc = r.get("pcie")) {
    for (const char* k : {"h2d", "d2h", "bidir", "kernel_read", "kernel_write"})
      if (const Json* v = pc->get(std::string(k) + "_gbps")) add("gpumon_diag_pcie_gbps", std::string("traffic=\"") + k + "\"", num(v));
    if (num(pc->get("link_gbps")) > 0) add("gpumon_diag_pcie_link_gbps", "", num(pc->get("link_gbps")));
  }
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
      if (kv.second.get("checked_in_kernel")) {
        add("gpumon_diag_gemm_checked_launches", "dtype=\"" + kv.first + "\",how=\"kernel\"", num(kv.second.get("checked_in_kernel")));
        add("gpumon_diag_gemm_checked_launches", "dtype=\"" + kv.first + "\",how=\"memory\"", num(kv.second.get("checked_in_memory")));
      }
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
}

// Node-wide members: collectives and the link matrix are about no single GPU.
void parse_fabric(const Json& r, Resolver& resolve, Families* f) {
  auto rccl = [&](const Json& sweep) {
    const int n = (int)num(sweep.get("gpus"), 0);
    if (const Json* pk = sweep.get("peak_busbw_gbps"))
      for (auto& kv : pk->obj)
        (*f)["gpumon_diag_rccl_busbw_gbps"][strprintf("gpus=\"%d\",op=\"%s\"", n, kv.first.c_str())] = num(&kv.second);
    if (const Json* ex = sweep.get("expected_busbw_gbps"))
      for (auto& kv : ex->obj)
        (*f)["gpumon_diag_rccl_expected_busbw_gbps"][strprintf("gpus=\"%d\",op=\"%s\"", n, kv.first.c_str())] = num(&kv.second);
  };
  if (const Json* s = r.get("rccl")) rccl(*s);
  if (const Json* s = r.get("rccl_scaling"))
    for (auto& e : s->arr) rccl(e);
  if (const Json* m = r.get("p2p_gbps"))
    for (size_t i = 0; i < m->arr.size(); i++)
      for (size_t j = 0; j < m->arr[i].arr.size(); j++)
        if (i != j) (*f)["gpumon_diag_p2p_gbps"][strprintf("src=\"%zu\",dst=\"%zu\"", i, j)] = num(&m->arr[i].arr[j]);
  if (const Json* l = r.get("links"))
    if (const Json* fs = l->get("findings"))
      for (auto& x : fs->arr) {
        const int idx = resolve(str(x.get("pci_bdf")), (int)num(x.get("gpu"), -1));
        if (idx < 0) continue;
        (*f)["gpumon_diag_link_findings"][strprintf("gpu=\"%d\",kind=\"%s\"", idx, str(x.get("kind")).c_str())] += 1;
      }
}

bool is_test_name(const std::string& s) {
  if (s.empty() || s.size() >= sizeof(((gm_diag_result_t*)nullptr)->test)) return false;
  for (char c : s)
    if (!isalnum((unsigned char)c) && c != '_' && c != '.' && c != '-') return false;
  return true;
}

// A whole report file.  `stem` names a node-wide report in its series.
void parse_report(const Json& r, const std::string& stem, const std::string& file, Resolver& resolve, CachedReport* c) {
  Families* f = &c->fams;
  const std::string scope = str(r.get("scope"));
  const bool node = scope == "node";
  const int report_gpu = r.get("gpu") ? (int)num(r.get("gpu")) : -1;
  const std::string report_bdf = str(r.get("pci_bdf"));
  if (node) {
    // pass / timestamp of a node-wide report under its own name, never under a GPU's
    if (const Json* res = r.get("result")) (*f)["gpumon_diag_node_pass"]["report=\"" + stem + "\""] = res->type == Json::STR && res->str != "fail" ? 1 : 0;
    if (const Json* ts = r.get("timestamp")) (*f)["gpumon_diag_node_timestamp_seconds"]["report=\"" + stem + "\""] = num(ts);
    if (const Json* gs = r.get("gpus"))
      if (gs->type == Json::ARR)
        for (auto& g : gs->arr) {
          if (g.type != Json::OBJ) continue;
          const int idx = resolve(str(g.get("pci_bdf")), (int)num(g.get("gpu"), -1));
          if (idx < 0) {
            GM_LOGW("exporter: %s: GPU %d (%s) is not one of the engine's; its results are not exported", file.c_str(), (int)num(g.get("gpu"), -1),
                    str(g.get("pci_bdf")).c_str());
            continue;
          }
          parse_gpu_report(g, idx, f);
        }
  } else {
    const int idx = resolve(report_bdf, report_gpu < 0 ? 0 : report_gpu);
    if (idx < 0) GM_LOGW("exporter: %s is about GPU %d (%s), which is not one of the engine's; not exported", file.c_str(), report_gpu, report_bdf.c_str());
    else parse_gpu_report(r, idx, f);
  }
  parse_fabric(r, resolve, f);

  // verdicts: the per-(GPU, test) health series, and what the engine is handed
  const Json* vs = r.get("verdicts");
  if (!vs || vs->type != Json::ARR) return;
  for (auto& v : vs->arr) {
    if (v.type != Json::OBJ) continue;
    const std::string test = str(v.get("test"));
    if (!is_test_name(test)) continue;
    Pending p;
    memset(&p.d, 0, sizeof p.d);
    p.d.version = GM_DIAG_RESULT_VERSION;
    // a multi-GPU test (the link test, a node run) names the GPU in each verdict; the others are about the report's GPU
    p.report_gpu = v.get("gpu") ? (int)num(v.get("gpu")) : report_gpu;
    p.pci_bdf = v.get("pci_bdf") ? str(v.get("pci_bdf")) : report_bdf;
    if (p.report_gpu < 0 && p.pci_bdf.empty()) continue;  // about the node as a whole: no GPU to charge
    copy_str(p.d.test, test);
    copy_str(p.d.message, str(v.get("message")));
    p.d.system = (uint32_t)num(v.get("system"), 0);
    p.d.health = (int32_t)num(v.get("health"), 0);
    const Json* val = v.get("value");
    p.d.value = val && val->type == Json::NUM ? val->num : std::nan("");
    p.d.ts_us = (int64_t)(num(r.get("timestamp"), 0) * 1e6);  // the same report read again is the same verdict
    const int idx = resolve(p.pci_bdf, p.report_gpu);
    if (idx >= 0) (*f)["gpumon_diag_test_health"][strprintf("gpu=\"%d\",test=\"%s\"", idx, test.c_str())] = p.d.health;
    c->verdicts.push_back(p);
  }
}

std::mutex g_mu;
std::map<std::string, CachedReport> g_cache;

// Hands the report's verdicts to the engine.  True when every one of them was accepted (or is about a card the
// engine does not have, which no retry will change).
bool submit_verdicts(CachedReport& c, const std::string& file, Resolver& resolve, Engine* engine) {
  bool ok = true;
  for (auto& p : c.verdicts) {
    const int idx = resolve(p.pci_bdf, p.report_gpu);
    if (idx < 0) {
      if (!c.said)
        GM_LOGW("exporter: %s: verdict %s is about GPU %d (%s), which is not one of the engine's; not submitted", file.c_str(), p.d.test, p.report_gpu,
                p.pci_bdf.c_str());
      continue;
    }
    gm_diag_result_t d = p.d;
    d.gpu = (uint32_t)idx;
    gm_status_t st = engine->diag_submit(d);
    if (st != GM_ST_OK) {
      if (!c.said) GM_LOGW("exporter: diagnostics verdict %s of GPU %u not accepted (%d); will be offered again", d.test, d.gpu, (int)st);
      ok = false;
    }
  }
  c.said = true;  // once per report: the retries that follow are silent
  return ok;
}

void load(const std::string& file, Families* merged, Engine* engine, Resolver& resolve) {
  struct stat st;
  if (stat(file.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) return;
  const int64_t mt = (int64_t)st.st_mtim.tv_sec * 1000000000ll + st.st_mtim.tv_nsec;
  CachedReport& c = g_cache[file];
  if (c.mtime_ns != mt) {
    c = CachedReport();
    c.mtime_ns = mt;
    std::string text;
    Json j;
    size_t slash = file.rfind('/'), dot = file.rfind('.');
    const std::string stem = file.substr(slash == std::string::npos ? 0 : slash + 1, dot == std::string::npos || (slash != std::string::npos && dot < slash) ? std::string::npos : dot - (slash == std::string::npos ? 0 : slash + 1));
    if (read_file(file, &text) && json_parse(text, &j) && j.type == Json::OBJ) parse_report(j, stem, file, resolve, &c);
    else GM_LOGW("exporter: %s is not a diagnostics report", file.c_str());
  }
  // Offered until accepted, and again once a minute after that: an engine that was away when the report appeared,
  // or that has restarted since and lost what it was told, ends up with the verdicts that are on disk.
  const int64_t now = monotonic_us();
  const bool due = c.submitted_ok ? now - c.submitted_mono_us >= 60000000 : !c.attempt_mono_us || now - c.attempt_mono_us >= 5000000;
  if (engine && !c.verdicts.empty() && due) {  // a refusal is retried every five seconds, not at every render
    c.attempt_mono_us = now;
    c.submitted_ok = submit_verdicts(c, file, resolve, engine);
    if (c.submitted_ok) c.submitted_mono_us = now;
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
  Resolver resolve{engine};
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
    for (auto& f : files) load(f, &merged, engine, resolve);
  } else {
    load(path, &merged, engine, resolve);
  }
  for (auto& kv : merged) {
    *out += strprintf("# HELP %s %s\n# TYPE %s gauge\n", kv.first.c_str(), help_of(kv.first), kv.first.c_str());
    // %.15g keeps Unix timestamps and byte counts exact.
    for (auto& s : kv.second) *out += strprintf("%s{%s} %.15g\n", kv.first.c_str(), s.first.c_str(), s.second);
  }
}

}  // namespace gpumon
