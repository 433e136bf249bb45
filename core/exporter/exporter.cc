#include "exporter.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <sys/stat.h>
#include <unistd.h>

#include "../common/json.h"
#include "../common/util.h"

namespace gpumon {
namespace {

struct Member {
  uint16_t field;
  const char* label;  // extra label name or nullptr
  std::string label_value;
  int link = -1;  // xGMI link index when the label is "link"
};
struct Family {
  std::string name, help;
  const char* type;
  std::vector<Member> members;
  bool last_not_idle = false;  // the synthesised gpu_last_not_idle_time family
  bool from_counters = false;  // fed by the hardware-counter plugin: sampled values in its child mode
};

const char* kBlockNames[] = {"umc", "gfx", "sdma", "mmhub", "xgmi"};

std::vector<Family> build_families(bool profiling, bool extras) {
  std::vector<Family> fams;
  auto add = [&](uint16_t id) {
    const FieldDef* f = field_def(id);
    if (!f || !f->prom_type) return;
    Family* fam = nullptr;
    for (auto& x : fams)
      if (x.name == f->prom_name) fam = &x;
    if (!fam) {
      fams.push_back(Family());
      fam = &fams.back();
      fam->name = f->prom_name;
      fam->help = f->prom_help;
      fam->type = f->prom_type == 2 ? "counter" : "gauge";
    }
    if (id == GM_FI_PROF_MFMA_ACTIVE || id == GM_FI_PROF_CU_ACTIVE || id == GM_FI_PROF_MEM_PIPE_ACTIVE) fam->from_counters = true;
    Member m;
    m.field = id;
    m.label = f->extra_label;
    if (f->extra_label)
      m.label_value = !strcmp(f->extra_label, "block") ? kBlockNames[f->extra_index] : strprintf("%d", f->extra_index);
    if (f->extra_label && !strcmp(f->extra_label, "link") && f->extra_index >= 0 && f->extra_index < GM_MAX_XGMI_LINKS) m.link = f->extra_index;
    fam->members.push_back(m);
  };
  for (uint16_t id : contract_fields()) {
    add(id);
    if (id == GM_FI_GFX_UTIL) {
      // Synthesised per GPU: wall-clock second latched whenever utilisation is
      // above 2 % or the GPU is seen for the first time
      // (exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:104-109,140).
      Family f;
      f.name = "dcgm_gpu_last_not_idle_time";
      f.help = "Timestamp of last time GPU utilization was 2% or less.";
      f.type = "gauge";
      f.last_not_idle = true;
      f.members.push_back(Member{GM_FI_GFX_UTIL, nullptr, ""});
      fams.push_back(f);
    }
  }
  if (profiling)
    for (uint16_t id : profiling_fields()) add(id);
  if (extras)
    for (uint16_t id : extra_fields()) add(id);
  return fams;
}

void append_escaped(std::string* out, const std::string& s) {
  for (char c : s) {
    if (c == '\\') *out += "\\\\";
    else if (c == '"') *out += "\\\"";
    else if (c == '\n') *out += "\\n";
    else *out += c;
  }
}

void append_value(std::string* out, const gm_value_t& v) {
  char buf[48];
  if (v.type == GM_FT_DOUBLE) {
    if (std::isnan(v.v.dbl)) { *out += "NaN"; return; }
    snprintf(buf, sizeof buf, "%.3f", v.v.dbl);
  } else {
    snprintf(buf, sizeof buf, "%lld", (long long)v.v.i64);
  }
  *out += buf;
}

}  // namespace

Exporter::Exporter(Engine* engine, const ExporterConfig& cfg) : engine_(engine), cfg_(cfg) {
  if (cfg_.interval_ms < 100) cfg_.interval_ms = 100;  // same floor as the reference's -d
  // The counter plugin's child mode delivers duty-cycle samples, not integrals: the HELP text says so, with the
  // numbers (three counter windows per probe run, one run per period).
  const char* mode = getenv("GPUMON_PROF_COUNTERS");
  if (mode && !strcmp(mode, "child")) {
    const char* pe = getenv("GPUMON_PROF_PERIOD_S");
    const char* we = getenv("GPUMON_PROF_WINDOW_MS");
    const double period = pe && atof(pe) > 0 ? atof(pe) : 30.0;
    const int window_ms = we && atoi(we) > 0 ? atoi(we) : 250;
    counter_note_ = strprintf(" Sampled: about %.2f s of counting every %.0f s.", 3 * window_ms / 1000.0, period);
  }
}

Exporter::~Exporter() {
  stop();
  save_state(true);
  if (policy_registered_) engine_->policy_unregister(policy_reg_);  // before the members the callback touches go away
  if (watching_) {
    engine_->unwatch_fields(group_, fg_);
    engine_->field_group_destroy(fg_);
    if (fg_slow_) {
      engine_->unwatch_fields(group_, fg_slow_);
      engine_->field_group_destroy(fg_slow_);
    }
    engine_->group_destroy(group_);
  }
}

gm_status_t Exporter::init() {
  std::vector<uint32_t> supported;
  gm_status_t st = engine_->supported_devices(&supported);
  if (st != GM_ST_OK) return st;
  gpus_.clear();
  std::string sel = trim(cfg_.gpus);
  if (sel.empty() || sel == "-1") {
    gpus_ = supported;
  } else {
    for (auto& t : split(sel, ',')) {
      std::string x = trim(t);
      if (x.empty()) continue;
      char* end = nullptr;
      long v = strtol(x.c_str(), &end, 10);
      if (*end || v < 0) return GM_ST_BADPARAM;
      if (std::find(supported.begin(), supported.end(), (uint32_t)v) == supported.end()) return GM_ST_GPU_NOT_SUPPORTED;
      if (std::find(gpus_.begin(), gpus_.end(), (uint32_t)v) == gpus_.end()) gpus_.push_back((uint32_t)v);
    }
    std::sort(gpus_.begin(), gpus_.end());
  }
  uuids_.clear();
  bdfs_.clear();
  partitions_.clear();
  link_peer_.clear();
  for (uint32_t g : gpus_) {
    gm_device_attributes_t a;
    a.version = GM_DEVICE_ATTRIBUTES_VERSION;
    st = engine_->device_attributes(g, &a);
    if (st != GM_ST_OK) return st;
    uuids_.push_back(a.uuid);
    std::string bdf = a.pci_bdf;
    for (auto& c : bdf) c = (char)tolower((unsigned char)c);
    bdfs_.push_back(bdf);
    // A partitioned GPU (DPX / QPX / CPX) is enumerated once per partition: the entries keep their own
    // gpu index and uuid, and say which slice of the device they are.
    int part = -1;
    if (a.compute_partition[0] && strcmp(a.compute_partition, "SPX") != 0) {
      const uint16_t fid = GM_FI_DEV_PARTITION_ID;
      gm_value_t v;
      if (engine_->latest_values(g, &fid, 1, &v) == GM_ST_OK && v.status == GM_VS_OK) part = (int)v.v.i64;
    }
    partitions_.push_back(part);
    // Who sits at the far end of each xGMI link: the `peer` label of the per-link series.
    std::vector<int> peers(GM_MAX_XGMI_LINKS, -1);
    std::vector<gm_xgmi_link_t> links;
    if (engine_->xgmi_links(g, &links) == GM_ST_OK)
      for (auto& l : links)
        if (l.link < GM_MAX_XGMI_LINKS) peers[l.link] = l.peer_gpu;
    link_peer_.push_back(peers);
  }
  fields_.clear();
  for (auto& fam : build_families(cfg_.profiling, cfg_.amd_extras))
    for (auto& m : fam.members)
      if (std::find(fields_.begin(), fields_.end(), m.field) == fields_.end()) fields_.push_back(m.field);

  static std::atomic<uint32_t> seq{0};
  std::string tag = strprintf("gpumon-exporter-%d-%u", (int)getpid(), seq++);
  st = engine_->group_create(tag, &group_);
  if (st != GM_ST_OK) return st;
  for (uint32_t g : gpus_)
    if ((st = engine_->group_add(group_, g)) != GM_ST_OK) return st;
  // Two schedules.  Almost every exported value comes out of the one bulk metrics table a GPU hands over per
  // tick; error counters, the bad-page table, performance level / power cap and memory usage are a device-library
  // call each and do not change at 10 Hz.  With a collection interval under slow_interval_ms those sources keep
  // the slower period, so a 100 ms exporter costs little more than the bulk read.
  std::vector<uint16_t> fast, slow;
  const uint32_t kSlowSources = SRC_ECC | SRC_BAD_PAGES | SRC_PERF | SRC_VRAM;
  const bool split = cfg_.slow_interval_ms > cfg_.interval_ms;
  for (uint16_t id : fields_) {
    const FieldDef* f = field_def(id);
    (split && f && (f->source & kSlowSources) ? slow : fast).push_back(id);
  }
  st = engine_->field_group_create(fast, tag, &fg_);
  if (st != GM_ST_OK) return st;
  st = engine_->watch_fields(group_, fg_, cfg_.interval_ms * 1000, GM_DEFAULT_MAX_KEEP_AGE_S, 2);
  if (st != GM_ST_OK) return st;
  if (!slow.empty()) {
    if ((st = engine_->field_group_create(slow, tag + "-slow", &fg_slow_)) != GM_ST_OK) return st;
    if ((st = engine_->watch_fields(group_, fg_slow_, cfg_.slow_interval_ms * 1000, GM_DEFAULT_MAX_KEEP_AGE_S, 2)) != GM_ST_OK) return st;
  }
  watching_ = true;
  if (cfg_.self_metrics) engine_->introspect_toggle(true);
  if (cfg_.health) {
    if ((st = engine_->health_set(group_, GM_HEALTH_WATCH_ALL)) != GM_ST_OK) return st;
    health_.assign(gpus_.size(), {});
    health_overall_.assign(gpus_.size(), GM_HEALTH_PASS);
    violations_.assign(gpus_.size(), {});
    st = engine_->policy_register(group_, GM_POLICY_COND_ALL, [this](const gm_violation_t& v) {
      auto it = std::find(gpus_.begin(), gpus_.end(), v.gpu);
      if (it == gpus_.end()) return;
      std::lock_guard<std::mutex> l(violation_mu_);
      for (int b = 0; b < kPolicyConditions; b++)
        if (v.condition == (1u << b)) violations_[(size_t)(it - gpus_.begin())][(size_t)b]++;
    }, &policy_reg_);
    if (st != GM_ST_OK) return st;
    policy_registered_ = true;
  }
  if (!cfg_.pod_map_path.empty() || !cfg_.kubelet_socket.empty()) refresh_pod_map();
  load_state();
  return GM_ST_OK;
}

// ---- state carried across restarts ----------------------------------------------
// Everything the GPUs count themselves (energy, ECC, violation accumulators) survives
// an exporter restart on its own.  The one value the exporter accumulates is the
// gpu_last_not_idle_time latch, which the reference's awk script starts at "now" on
// every start (dcgm-exporter:97-113): an idle-GPU reaper that reads it sees a GPU
// "become busy" whenever systemd restarts the exporter.  With a state path the latch
// is kept per GPU UUID, written atomically at most every 10 s and at shutdown.
void Exporter::load_state() {
  if (cfg_.state_path.empty()) return;
  std::string text;
  Json j;
  if (!read_file(cfg_.state_path, &text)) return;  // first start
  if (!json_parse(text, &j) || j.type != Json::OBJ) {
    GM_LOGW("exporter: ignoring unreadable state file %s", cfg_.state_path.c_str());
    return;
  }
  const Json* latch = j.get("last_not_idle");
  if (!latch || latch->type != Json::OBJ) return;
  const int64_t now_s = now_us() / 1000000;
  std::lock_guard<std::mutex> l(mu_);
  for (size_t g = 0; g < gpus_.size(); g++) {
    const Json* v = latch->get(uuids_[g]);
    if (!v || v->type != Json::NUM) continue;
    const int64_t ts = (int64_t)v->num;
    if (ts > 0 && ts <= now_s) last_not_idle_[gpus_[g]] = ts;  // a timestamp from the future is a clock step: start over
  }
}

void Exporter::save_state(bool force) {
  if (cfg_.state_path.empty()) return;
  std::string text;
  {
    std::lock_guard<std::mutex> l(mu_);
    const int64_t now = monotonic_us();
    if (!state_dirty_ || (!force && state_saved_mono_ && now - state_saved_mono_ < 10000000)) return;
    text = "{\"version\":1,\"last_not_idle\":{";
    bool first = true;
    for (size_t g = 0; g < gpus_.size(); g++) {
      auto it = last_not_idle_.find(gpus_[g]);
      if (it == last_not_idle_.end() || uuids_[g].empty()) continue;
      text += strprintf("%s\"%s\":%lld", first ? "" : ",", json_escape(uuids_[g]).c_str(), (long long)it->second);
      first = false;
    }
    text += "}}\n";
    state_dirty_ = false;
    state_saved_mono_ = now;
  }
  if (!write_file_atomic(cfg_.state_path, text.data(), text.size()))
    GM_LOGW("exporter: cannot write state file %s: %s", cfg_.state_path.c_str(), strerror(errno));
}

void Exporter::set_pod_map(const PodMap& m) {
  std::lock_guard<std::mutex> l(mu_);
  pod_map_ = m;
}

// Runs on the caller's thread at start-up and on the pod thread afterwards, never on the publish thread: a kubelet
// that does not answer holds up nothing but the next pod map.
void Exporter::refresh_pod_map() {
  PodMap m;
  bool ok = false;
  if (!cfg_.kubelet_socket.empty()) {
    std::string err;
    if (!kubelet_) kubelet_.reset(new KubeletClient(cfg_.kubelet_socket, cfg_.resource_names));
    gm_status_t st = kubelet_->list(10000, &m, &err);
    kubelet_connects_ = kubelet_->connects();
    ok = st == GM_ST_OK;
    // Said when it starts failing, once a minute while it keeps failing, and when it works again: a kubelet that is
    // down for an hour is two lines and sixty reminders, not one line per query.
    const int64_t now = monotonic_us();
    if (!ok && (!kubelet_failing_ || now - kubelet_warned_mono_ > 60 * 1000000ll)) {
      GM_LOGW("pod attribution: kubelet query failed: %s%s", err.c_str(), kubelet_failing_ ? " (still)" : "");
      kubelet_warned_mono_ = now;
    }
    if (ok && kubelet_failing_) GM_LOGI("pod attribution: the kubelet answers again");
    kubelet_failing_ = !ok;
  }
  if (!ok && !cfg_.pod_map_path.empty()) {
    std::string text;
    if (read_file(cfg_.pod_map_path, &text)) {
      if (!text.empty() && text[0] == '{') ok = parse_device_plugin_checkpoint(text, cfg_.resource_names, &m);
      else { m = parse_pod_map_tsv(text); ok = true; }
    }
  }
  pod_map_ok_ = ok ? 1 : 0;
  pod_refreshes_++;
  last_pod_refresh_mono_ = monotonic_us();
  if (ok) set_pod_map(m);  // on failure keep the last good map
}

void Exporter::render(std::string* out) {
  int64_t t0 = monotonic_us();
  static const std::vector<Family> fam_all[4] = {build_families(false, false), build_families(true, false),
                                                 build_families(false, true), build_families(true, true)};
  const std::vector<Family>& fams = fam_all[(cfg_.profiling ? 1 : 0) | (cfg_.amd_extras ? 2 : 0)];
  const size_t nf = fields_.size(), ng = gpus_.size();
  std::lock_guard<std::mutex> l(mu_);
  scratch_.resize(nf * ng);
  std::vector<bool> gpu_ok(ng, false);
  for (size_t g = 0; g < ng; g++)
    gpu_ok[g] = engine_->latest_values(gpus_[g], fields_.data(), (int)nf, &scratch_[g * nf]) == GM_ST_OK;
  auto value_of = [&](size_t g, uint16_t field) -> const gm_value_t* {
    for (size_t i = 0; i < nf; i++)
      if (fields_[i] == field) return &scratch_[g * nf + i];
    return nullptr;
  };
  int64_t now_s = now_us() / 1000000;

  // Label blocks are fixed per GPU for one render.
  std::vector<std::string> base(ng), podl(ng);
  for (size_t g = 0; g < ng; g++) {
    base[g] = strprintf("{gpu=\"%u\",uuid=\"", gpus_[g]);
    append_escaped(&base[g], uuids_[g]);
    base[g] += "\"";
    if (partitions_[g] >= 0) base[g] += strprintf(",partition=\"%d\"", partitions_[g]);
    const PodInfo* p = nullptr;
    auto it = pod_map_.find(uuids_[g]);
    if (it != pod_map_.end()) p = &it->second;
    if (!p && (it = pod_map_.find(bdfs_[g])) != pod_map_.end()) p = &it->second;
    static const char* prefixes[] = {"amdgpu", "gpu", "card", "nvidia"};
    for (size_t k = 0; !p && k < sizeof prefixes / sizeof prefixes[0]; k++)
      if ((it = pod_map_.find(strprintf("%s%u", prefixes[k], gpus_[g]))) != pod_map_.end()) p = &it->second;
    if (p) {
      podl[g] = ",pod_name=\"";
      append_escaped(&podl[g], p->pod);
      podl[g] += "\",pod_namespace=\"";
      append_escaped(&podl[g], p->ns);
      podl[g] += "\",container_name=\"";
      append_escaped(&podl[g], p->container);
      podl[g] += "\"";
    }
  }

  out->clear();
  out->reserve(ng * fams.size() * 96 + 4096);
  auto emit_sample = [&](const Family& fam, const Member& m, size_t g) {
    if (!gpu_ok[g]) return false;
    const gm_value_t* v = value_of(g, m.field);
    if (!v || v->status != GM_VS_OK) return false;  // blank values are omitted, like "N/A" in the reference
    *out += fam.name;
    *out += base[g];
    if (m.label) {
      *out += ",";
      *out += m.label;
      *out += "=\"";
      *out += m.label_value;
      *out += "\"";
      if (m.link >= 0 && link_peer_[g][(size_t)m.link] >= 0) *out += strprintf(",peer=\"%d\"", link_peer_[g][(size_t)m.link]);
    }
    *out += podl[g];
    *out += "} ";
    if (fam.last_not_idle) {
      int64_t util = v->type == GM_FT_DOUBLE ? (int64_t)v->v.dbl : v->v.i64;
      auto it = last_not_idle_.find(gpus_[g]);
      if (it == last_not_idle_.end() || util > 2) {
        if (it == last_not_idle_.end() || it->second != now_s) state_dirty_ = true;
        last_not_idle_[gpus_[g]] = now_s;
      }
      *out += strprintf("%lld", (long long)last_not_idle_[gpus_[g]]);
    } else {
      append_value(out, *v);
    }
    *out += "\n";
    return true;
  };
  auto emit_header = [&](const Family& fam) {
    *out += "# HELP ";
    *out += fam.name;
    *out += " ";
    *out += fam.help;
    if (fam.from_counters) *out += counter_note_;
    *out += "\n# TYPE ";
    *out += fam.name;
    *out += " ";
    *out += fam.type;
    *out += "\n";
  };

  if (!cfg_.legacy_layout) {
    // Valid exposition format: HELP, TYPE, then every sample of the family.
    for (auto& fam : fams) {
      size_t mark = out->size();
      emit_header(fam);
      size_t after_header = out->size();
      for (size_t g = 0; g < ng; g++)
        for (auto& m : fam.members) emit_sample(fam, m, g);
      if (out->size() == after_header) out->resize(mark);  // nothing to say: no orphan header
    }
  } else {
    // GPU-major layout of the reference's awk program: headers accompany the
    // first exported GPU only (dcgm-exporter:97-113).
    for (size_t g = 0; g < ng; g++)
      for (auto& fam : fams)
        for (auto& m : fam.members) {
          size_t mark = out->size();
          if (g == 0) emit_header(fam);
          size_t after_header = out->size();
          if (!emit_sample(fam, m, g) && g == 0 && out->size() == after_header) out->resize(mark);
        }
  }
  if (cfg_.health) {
    // A check that could not run (the engine went away) must not look like a clean bill of health: the verdict
    // families are left out and this gauge says why.
    *out += strprintf("# HELP gpumon_health_check_up 1 if the last health check ran, 0 if the engine refused or could not be reached.\n"
                      "# TYPE gpumon_health_check_up gauge\ngpumon_health_check_up %d\n", health_valid_ ? 1 : 0);
    if (health_valid_) render_health(base, podl, out);
  }
  if (cfg_.self_metrics) render_self(out);
  if (!cfg_.diag_results_path.empty()) render_diag_results(cfg_.diag_results_path, out, engine_);
  last_render_us_ = monotonic_us() - t0;
}

// One health check per sweep (not per scrape: the checks that look at the growth of a counter compare with the
// previous check, and a scraper must not shorten that window).
void Exporter::check_health() {
  std::vector<gm_health_response_t> rs;
  gm_status_t st = engine_->health_check(group_, &rs);
  std::lock_guard<std::mutex> l(mu_);
  if (st != GM_ST_OK) {
    health_valid_ = false;
    return;
  }
  for (auto& r : rs) {
    auto it = std::find(gpus_.begin(), gpus_.end(), r.gpu);
    if (it == gpus_.end()) continue;
    const size_t g = (size_t)(it - gpus_.begin());
    for (auto& c : health_[g]) c.level = GM_HEALTH_PASS;
    health_overall_[g] = r.overall;
    for (uint32_t i = 0; i < r.incident_count && i < GM_MAX_INCIDENTS; i++) {
      const gm_incident_t& in = r.incidents[i];
      for (int b = 0; b < kHealthSystems; b++) {
        if (in.system != (1u << b)) continue;
        HealthCell& c = health_[g][(size_t)b];
        c.level = std::max(c.level, in.health);
        if (in.health >= GM_HEALTH_FAIL) c.failures++;
        else if (in.health >= GM_HEALTH_WARN) c.warnings++;
      }
    }
  }
  health_valid_ = true;
}

void Exporter::render_health(const std::vector<std::string>& base, const std::vector<std::string>& podl, std::string* out) {
  static const char* kSystem[kHealthSystems] = {"pcie", "xgmi", "memory", "compute", "thermal", "power", "driver", "firmware"};
  *out += "# HELP gpumon_health_status Result of the engine's health check per GPU subsystem (0 = pass, 10 = warning, 20 = failure).\n"
          "# TYPE gpumon_health_status gauge\n";
  for (size_t g = 0; g < gpus_.size(); g++)
    for (int b = 0; b < kHealthSystems; b++)
      *out += strprintf("gpumon_health_status%s,system=\"%s\"%s} %d\n", base[g].c_str(), kSystem[b], podl[g].c_str(), health_[g][(size_t)b].level);
  *out += "# HELP gpumon_health_overall Worst subsystem result of the GPU's last health check (0 = pass, 10 = warning, 20 = failure).\n"
          "# TYPE gpumon_health_overall gauge\n";
  for (size_t g = 0; g < gpus_.size(); g++)
    *out += strprintf("gpumon_health_overall%s%s} %d\n", base[g].c_str(), podl[g].c_str(), health_overall_[g]);
  *out += "# HELP gpumon_health_incidents_total Health incidents reported since the exporter started, by subsystem and level.\n"
          "# TYPE gpumon_health_incidents_total counter\n";
  for (size_t g = 0; g < gpus_.size(); g++)
    for (int b = 0; b < kHealthSystems; b++) {
      const HealthCell& c = health_[g][(size_t)b];
      *out += strprintf("gpumon_health_incidents_total%s,system=\"%s\",level=\"warning\"%s} %llu\n", base[g].c_str(), kSystem[b],
                        podl[g].c_str(), (unsigned long long)c.warnings);
      *out += strprintf("gpumon_health_incidents_total%s,system=\"%s\",level=\"failure\"%s} %llu\n", base[g].c_str(), kSystem[b],
                        podl[g].c_str(), (unsigned long long)c.failures);
    }
  static const char* kCondition[kPolicyConditions] = {"ecc_ue", "pcie", "bad_pages", "thermal", "power", "xgmi", "gpu_event"};
  *out += "# HELP gpumon_policy_violations_total Policy violations since the exporter started, by condition.\n"
          "# TYPE gpumon_policy_violations_total counter\n";
  std::lock_guard<std::mutex> l(violation_mu_);
  for (size_t g = 0; g < gpus_.size(); g++)
    for (int b = 0; b < kPolicyConditions; b++)
      *out += strprintf("gpumon_policy_violations_total%s,condition=\"%s\"%s} %llu\n", base[g].c_str(), kCondition[b], podl[g].c_str(),
                        (unsigned long long)violations_[g][(size_t)b]);
}

void Exporter::render_self(std::string* out) {
  auto gauge = [&](const char* name, const char* help, double v, const char* type = "gauge") {
    *out += strprintf("# HELP %s %s\n# TYPE %s %s\n%s %.6g\n", name, help, name, type, name, v);
  };
  gm_introspect_t in;
  in.version = GM_INTROSPECT_VERSION;
  if (engine_->introspect(&in) == GM_ST_OK) {
    gauge("gpumon_engine_memory_kb", "Resident memory of the host engine (in KB).", (double)in.memory_kb);
    gauge("gpumon_engine_cpu_percent", "CPU utilization of the host engine (in % of one core).", in.cpu_percent);
    gauge("gpumon_engine_sample_cycle_seconds", "Wall time of the last sample cycle over all GPUs.", in.last_cycle_ms / 1e3);
    gauge("gpumon_engine_sample_cycles_total", "Sample cycles completed.", (double)in.sample_cycles, "counter");
    gauge("gpumon_engine_backend_errors_total", "Device-library calls that failed.", (double)in.backend_errors, "counter");
  }
  // Where each GPU's process list comes from on this host; "open-files" means per-process memory is unknown.
  // Asking costs a look at the process list (a walk over /proc where the driver's table is closed): once a minute,
  // not once a scrape.
  const int64_t now_mono = monotonic_us();
  if (proc_source_.size() != gpus_.size() || now_mono - proc_source_mono_ > 60 * 1000000ll) {
    proc_source_.assign(gpus_.size(), "");
    for (size_t i = 0; i < gpus_.size(); i++)
      if (engine_->process_source(gpus_[i], &proc_source_[i]) != GM_ST_OK) proc_source_[i].clear();
    proc_source_mono_ = now_mono;
  }
  std::string src_lines;
  for (size_t i = 0; i < gpus_.size(); i++)
    if (!proc_source_[i].empty())
      src_lines += strprintf("gpumon_engine_process_source{gpu=\"%u\",source=\"%s\"} 1\n", gpus_[i], proc_source_[i].c_str());
  if (!src_lines.empty())
    *out += "# HELP gpumon_engine_process_source Source of the per-GPU process list: kfd, fdinfo, open-files, fake or none.\n"
            "# TYPE gpumon_engine_process_source gauge\n" + src_lines;
  // Whether the bulk metrics table is read from the driver's file directly (1) or through the device library (0).
  std::string table_lines;
  for (uint32_t g : gpus_) {
    std::string src;
    if (engine_->metrics_table_source(g, &src) == GM_ST_OK && src != "n/a")
      table_lines += strprintf("gpumon_engine_metrics_table_direct{gpu=\"%u\"} %d\n", g, src == "direct" ? 1 : 0);
  }
  if (!table_lines.empty())
    *out += "# HELP gpumon_engine_metrics_table_direct 1 if the bulk metrics table is read from the driver's file directly, 0 if through the device library.\n"
            "# TYPE gpumon_engine_metrics_table_direct gauge\n" + table_lines;
  gauge("gpumon_exporter_render_duration_seconds", "Wall time of the previous text render.", (double)last_render_us_ / 1e6);
  gauge("gpumon_exporter_sweeps_total", "File sweeps published.", (double)sweeps_, "counter");
  gauge("gpumon_exporter_scrapes_total", "HTTP scrapes served.", (double)scrapes_, "counter");
  gauge("gpumon_exporter_publish_errors_total", "File publishes that failed.", (double)publish_errors_, "counter");
  if (pod_map_ok_ >= 0) {
    gauge("gpumon_exporter_pod_map_up", "1 if the last pod-resources query succeeded.", (double)pod_map_ok_);
    gauge("gpumon_exporter_pod_map_refreshes_total", "Pod-map queries made (kubelet or file).", (double)pod_refreshes_, "counter");
    if (!cfg_.kubelet_socket.empty())
      gauge("gpumon_exporter_kubelet_connects_total", "Connections opened to the kubelet (one is kept between queries).", (double)kubelet_connects_, "counter");
  }
}

// What a sweep without an output file still has to do: scrapes render on demand, but the last-not-idle latch must
// see every sample cycle, not only the ones a scraper happens to look at.
void Exporter::update_latch() {
  std::lock_guard<std::mutex> l(mu_);
  const int64_t now_s = now_us() / 1000000;
  const uint16_t field = GM_FI_GFX_UTIL;
  for (uint32_t g : gpus_) {
    gm_value_t v;
    if (engine_->latest_values(g, &field, 1, &v) != GM_ST_OK || v.status != GM_VS_OK) continue;
    const int64_t util = v.type == GM_FT_DOUBLE ? (int64_t)v.v.dbl : v.v.i64;
    auto it = last_not_idle_.find(g);
    if (it == last_not_idle_.end() || util > 2) {
      if (it == last_not_idle_.end() || it->second != now_s) state_dirty_ = true;
      last_not_idle_[g] = now_s;
    }
  }
}

gm_status_t Exporter::publish() {
  if (cfg_.health) check_health();
  if (cfg_.output_path.empty()) {
    update_latch();
    // Reports of the diagnostics are read, and their verdicts handed to the engine, when the text is rendered; with
    // no file to write that would wait for the next scrape.  Once a second is soon enough for a health incident.
    const int64_t now = monotonic_us();
    if (!cfg_.diag_results_path.empty() && now - diag_polled_mono_ >= 1000000) {
      diag_polled_mono_ = now;
      std::string unused;
      render_diag_results(cfg_.diag_results_path, &unused, engine_);
    }
  } else {
    std::string text;
    render(&text);
    size_t slash = cfg_.output_path.rfind('/');
    if (slash != std::string::npos && slash > 0) mkdir(cfg_.output_path.substr(0, slash).c_str(), 0755);
    if (!write_file_atomic(cfg_.output_path, text.data(), text.size())) {
      publish_errors_++;
      GM_LOGE("exporter: cannot publish %s: %s", cfg_.output_path.c_str(), strerror(errno));
      return GM_ST_GENERIC_ERROR;
    }
  }
  sweeps_++;
  last_sweep_us_ = now_us();
  save_state(false);
  return GM_ST_OK;
}

gm_status_t Exporter::sweep() {
  gm_status_t st = engine_->update_all(true);
  if (st != GM_ST_OK) return st;
  return publish();
}

void Exporter::handle_http(const HttpRequest& req, HttpResponse* resp) {
  if (req.path == "/metrics" || req.path == "/gpu/metrics") {
    scrapes_++;
    render(&resp->body);
    resp->content_type = "text/plain; version=0.0.4; charset=utf-8";
    return;
  }
  if (req.path == "/healthz") {
    // Live while sweeps keep coming (the reference's pod exporter dies after
    // 10 minutes without a new file: src/watchers.go:56-59).
    int64_t age_us = now_us() - last_sweep_us_.load();
    bool ok = !running_ || last_sweep_us_ == 0 || age_us < std::max<int64_t>(600000000, cfg_.interval_ms * 3000);
    resp->status = ok ? 200 : 503;
    resp->body = ok ? "ok\n" : "stale\n";
    return;
  }
  if (cfg_.rest_api && handle_rest(engine_, req, resp)) return;
  resp->status = 404;
  resp->body = "404 page not found\n";
}

void Exporter::loop() {
  // The first sweep forces a sample, which also anchors the sampler's drift-free
  // schedule; later iterations only render what the sampler has cached.  Both
  // schedules run off the monotonic clock, so publishing a fixed margin after
  // every sampler tick keeps the file one tick fresh without sampling twice.
  int64_t anchor = monotonic_us();
  gm_status_t st = sweep();
  if (st != GM_ST_OK) GM_LOGW("exporter sweep failed: %s", status_string(st));
  const int64_t interval = cfg_.interval_ms * 1000;
  int64_t next = anchor + interval + std::min<int64_t>(interval / 10, 100000);
  while (running_) {
    {
      std::unique_lock<std::mutex> l(wake_mu_);
      int64_t now = monotonic_us();
      if (next > now) cv_wait_for(wake_cv_, l, std::chrono::microseconds(next - now), [&] { return !running_.load(); });
    }
    if (!running_) break;
    if ((st = publish()) != GM_ST_OK) GM_LOGW("exporter publish failed: %s", status_string(st));
    next += interval;
    int64_t now = monotonic_us();
    if (next <= now) next = now + interval;
  }
}

// The pod map on a schedule and a thread of its own: one kept connection to the kubelet, asked every
// pod_refresh_ms (the reference dials anew once a second, on the path that publishes: src/watchers.go:41).
void Exporter::pod_loop() {
  const int64_t period = std::max<int64_t>(cfg_.pod_refresh_ms, 100) * 1000;
  while (running_) {
    {
      std::unique_lock<std::mutex> l(wake_mu_);
      const int64_t due = last_pod_refresh_mono_ + period, now = monotonic_us();
      if (due > now) cv_wait_for(wake_cv_, l, std::chrono::microseconds(due - now), [&] { return !running_.load(); });
    }
    if (!running_) break;
    refresh_pod_map();
  }
}

gm_status_t Exporter::start() {
  if (running_.exchange(true)) return GM_ST_OK;
  if (!cfg_.listen.empty()) {
    if (!http_.start(cfg_.listen, [this](const HttpRequest& rq, HttpResponse* rs) { handle_http(rq, rs); })) {
      running_ = false;
      return GM_ST_GENERIC_ERROR;
    }
  }
  th_ = std::thread([this] { loop(); });
  if (!cfg_.kubelet_socket.empty() || !cfg_.pod_map_path.empty()) pod_th_ = std::thread([this] { pod_loop(); });
  return GM_ST_OK;
}

void Exporter::stop() {
  if (!running_.exchange(false)) return;
  wake_cv_.notify_all();
  if (th_.joinable()) th_.join();
  if (pod_th_.joinable()) pod_th_.join();
  http_.stop();
}

}  // namespace gpumon
