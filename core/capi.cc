// C ABI glue: the process-global, reference-counted engine handle and one thin
// wrapper per gpumon.h entry point.
#include <csignal>
#include <dlfcn.h>
#include <cstring>
#include <fcntl.h>
#include <spawn.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

extern char** environ;

#include "backend/metrics_table.h"
#include "backend/procscan.h"
#include "common/util.h"
#include "engine/engine.h"
#include "exporter/exporter.h"
#include "exporter/pod_exporter.h"
#include "exporter/views.h"
#include "gpumon.h"
#include "rpc/server.h"

using namespace gpumon;

namespace {

struct PolicyQueue {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<gm_violation_t> q;
  gm_violation_cb cb = nullptr;
  void* user = nullptr;
};

struct EventSet {
  std::mutex mu;
  std::condition_variable cv;
  std::deque<gm_event_t> q;
  uint64_t mask[GM_MAX_GPUS + 1];  // per GPU; index GM_MAX_GPUS = all GPUs
  bool registered = false;
  uint32_t sub = 0;
};

struct Global {
  std::mutex mu;  // lifecycle only
  int refcount = 0;
  int mode = GM_MODE_EMBEDDED;
  std::unique_ptr<LocalEngine> local;
  std::unique_ptr<Engine> remote;
  Engine* engine = nullptr;
  std::unique_ptr<RpcServer> server;
  pid_t child = -1;
  std::string child_socket;

  std::mutex obj_mu;
  std::map<gm_policy_reg_t, std::shared_ptr<PolicyQueue>> policy_queues;
  std::map<gm_event_set_t, std::shared_ptr<EventSet>> event_sets;
  std::map<gm_exporter_t, std::shared_ptr<Exporter>> exporters;
  uint32_t next_es = 1, next_exp = 1;
};

Global& G() {
  static Global* g = new Global();  // intentionally leaked: threads may outlive static destructors
  return *g;
}

Engine* engine() { return G().engine; }

#define REQUIRE_ENGINE()            \
  Engine* e = engine();             \
  if (!e) return GM_ST_UNINITIALIZED

std::string find_hostengine(const char* hint) {
  if (hint && *hint) return hint;
  const char* env = getenv("GPUMON_HOSTENGINE");
  if (env && *env) return env;
  Dl_info info;
  if (dladdr((void*)&find_hostengine, &info) && info.dli_fname) {
    std::string dir = info.dli_fname;
    size_t s = dir.rfind('/');
    dir = s == std::string::npos ? "." : dir.substr(0, s);
    for (const char* rel : {"/gpumon-hostengine", "/../bin/gpumon-hostengine", "/bin/gpumon-hostengine"}) {
      std::string p = dir + rel;
      if (access(p.c_str(), X_OK) == 0) return p;
    }
  }
  const char* path = getenv("PATH");
  if (path)
    for (auto& d : split(path, ':')) {
      std::string p = d + "/gpumon-hostengine";
      if (access(p.c_str(), X_OK) == 0) return p;
    }
  return std::string();
}

void stop_child(Global& g) {
  // Polite stop with a 10 s grace, then SIGKILL — the same escalation as the
  // reference's exporter script (dcgm-exporter:198-217).
  if (g.child > 0) kill(g.child, SIGTERM);
  for (int i = 0; i < 100 && g.child > 0; i++) {
    int status;
    if (waitpid(g.child, &status, WNOHANG) == g.child) {
      g.child = -1;
      break;
    }
    usleep(100000);
  }
  if (g.child > 0) {
    kill(g.child, SIGKILL);
    waitpid(g.child, nullptr, 0);
    g.child = -1;
  }
  if (!g.child_socket.empty()) {
    unlink(g.child_socket.c_str());
    const size_t slash = g.child_socket.rfind('/');
    if (slash != std::string::npos && slash > 0) rmdir(g.child_socket.substr(0, slash).c_str());  // its private directory
  }
  g.child_socket.clear();
}

gm_status_t do_init(Global& g, const gm_init_params_t* p) {
  gm_init_params_t def;
  memset(&def, 0, sizeof def);
  def.version = GM_INIT_PARAMS_VERSION;
  if (!p) p = &def;
  if (p->version != GM_INIT_PARAMS_VERSION) return GM_ST_VER_MISMATCH;
  g.mode = p->mode;
  const char* env_backend = getenv("GPUMON_BACKEND");
  const char* env_spec = getenv("GPUMON_BACKEND_SPEC");
  std::string backend = p->backend ? p->backend : (env_backend ? env_backend : "amdsmi");
  std::string spec = p->backend_spec ? p->backend_spec : (env_spec ? env_spec : "");

  if (p->mode == GM_MODE_EMBEDDED) {
    std::unique_ptr<Backend> b = make_backend(backend, spec);
    if (!b) return GM_ST_BADPARAM;
    g.local.reset(new LocalEngine(std::move(b)));
    gm_status_t st = g.local->start();
    if (st != GM_ST_OK) {
      g.local.reset();
      return st;
    }
    g.engine = g.local.get();
    return GM_ST_OK;
  }

  std::string address;
  bool is_unix = false;
  if (p->mode == GM_MODE_STANDALONE) {
    if (!p->address || !*p->address) return GM_ST_BADPARAM;
    address = p->address;
    is_unix = p->address_is_unix_socket != 0;
  } else if (p->mode == GM_MODE_START_HOSTENGINE) {
    std::string bin = find_hostengine(p->hostengine_path);
    if (bin.empty()) {
      GM_LOGE("gpumon-hostengine not found (set GPUMON_HOSTENGINE or PATH)");
      return GM_ST_CHILD_SPAWN_FAILED;
    }
    // The socket lives in a directory of its own that only this user can enter (mkdtemp creates it 0700), so the
    // name cannot be taken by someone else between its choice and the daemon's bind.
    char tmpl[] = "/tmp/gpumon-XXXXXX";
    if (!mkdtemp(tmpl)) return GM_ST_CHILD_SPAWN_FAILED;
    address = std::string(tmpl) + "/hostengine.sock";
    is_unix = true;
    // posix_spawn, not fork: the caller is an arbitrary, usually multithreaded host (Python with a GPU runtime, a Go
    // program), and a forked copy of such a process must not run anything but exec.  Own process group: terminal
    // signals to the parent do not reach the daemon.
    for (const char* hip : {"libamdhip64.so", "libamdhip64.so.7", "libamdhip64.so.6"})
      if (void* h = dlopen(hip, RTLD_LAZY | RTLD_NOLOAD)) {
        dlclose(h);
        // Worth a line in the log: on hosts that forbid replacing a GPU-initialised program this is the call to look at.
        GM_LOGW("starting gpumon-hostengine from a process that has the HIP runtime loaded (posix_spawn: no copy of this process runs)");
        break;
      }
    std::vector<std::string> args = {bin, "--domain-socket", address, "--backend", backend};
    if (!spec.empty()) {
      args.push_back("--backend-spec");
      args.push_back(spec);
    }
    std::vector<char*> argv;
    for (auto& a : args) argv.push_back(const_cast<char*>(a.c_str()));
    argv.push_back(nullptr);
    posix_spawnattr_t attr;
    posix_spawnattr_init(&attr);
    posix_spawnattr_setflags(&attr, POSIX_SPAWN_SETPGROUP);
    posix_spawnattr_setpgroup(&attr, 0);
    pid_t pid = -1;
    const int rc = posix_spawn(&pid, bin.c_str(), nullptr, &attr, argv.data(), environ);
    posix_spawnattr_destroy(&attr);
    if (rc != 0) {
      GM_LOGE("cannot start %s: %s", bin.c_str(), strerror(rc));
      rmdir(tmpl);
      return GM_ST_CHILD_SPAWN_FAILED;
    }
    g.child = pid;
    g.child_socket = address;
  } else {
    return GM_ST_BADPARAM;
  }
  gm_status_t st;
  std::string err;
  if (p->mode == GM_MODE_START_HOSTENGINE) {
    // Short connect attempts while the child comes up, so that a child that died
    // (bad binary, backend init failure) is reported at once instead of after the
    // whole grace period.
    st = GM_ST_CONNECTION_NOT_VALID;
    for (int64_t t0 = monotonic_us(); monotonic_us() - t0 < 15000000 && !g.remote;) {
      int status;
      if (waitpid(g.child, &status, WNOHANG) == g.child) {
        err = strprintf("host engine child exited early (status 0x%x)", status);
        g.child = -1;
        st = GM_ST_CHILD_SPAWN_FAILED;
        break;
      }
      g.remote = make_remote_engine(address, is_unix, 200, &st, &err);
    }
  } else {
    g.remote = make_remote_engine(address, is_unix, 5000, &st, &err);
  }
  if (!g.remote) {
    GM_LOGE("%s", err.c_str());
    stop_child(g);
    return st;
  }
  g.engine = g.remote.get();
  return GM_ST_OK;
}

void do_shutdown(Global& g) {
  {
    std::lock_guard<std::mutex> l(g.obj_mu);
    for (auto& kv : g.exporters) kv.second->stop();
    g.exporters.clear();
    g.policy_queues.clear();
    g.event_sets.clear();
  }
  if (g.server) {
    g.server->stop();
    g.server.reset();
  }
  g.engine = nullptr;
  g.remote.reset();
  if (g.local) {
    g.local->stop();
    g.local.reset();
  }
  stop_child(g);
}

}  // namespace

extern "C" {

const char* gm_strerror(gm_status_t st) { return status_string(st); }
void gm_set_log_level(int level) { set_log_level(level); }

gm_status_t gm_init(const gm_init_params_t* params) {
  Global& g = G();
  std::lock_guard<std::mutex> l(g.mu);
  if (g.refcount < 0) return GM_ST_GENERIC_ERROR;
  if (g.refcount == 0) {
    gm_status_t st = do_init(g, params);
    if (st != GM_ST_OK) return st;
  }
  g.refcount++;
  return GM_ST_OK;
}

gm_status_t gm_shutdown(void) {
  Global& g = G();
  std::lock_guard<std::mutex> l(g.mu);
  if (g.refcount <= 0) return GM_ST_UNINITIALIZED;
  if (g.refcount == 1) do_shutdown(g);
  g.refcount--;
  return GM_ST_OK;
}

int gm_init_refcount(void) {
  Global& g = G();
  std::lock_guard<std::mutex> l(g.mu);
  return g.refcount;
}

gm_status_t gm_server_start(const char* address, int is_unix) {
  Global& g = G();
  std::lock_guard<std::mutex> l(g.mu);
  if (!g.engine) return GM_ST_UNINITIALIZED;
  if (!address) return GM_ST_BADPARAM;
  if (g.server) return GM_ST_IN_USE;
  g.server.reset(new RpcServer(g.engine));
  gm_status_t st = g.server->start(address, is_unix != 0);
  if (st != GM_ST_OK) g.server.reset();
  return st;
}

gm_status_t gm_server_stop(void) {
  Global& g = G();
  std::lock_guard<std::mutex> l(g.mu);
  if (!g.server) return GM_ST_NOT_CONFIGURED;
  g.server->stop();
  g.server.reset();
  return GM_ST_OK;
}

// ------------------------------------------------------------- field meta --
gm_status_t gm_field_meta(uint16_t field_id, gm_field_meta_t* out) {
  const FieldDef* f = field_def(field_id);
  if (!f || !out) return GM_ST_NOT_FOUND;
  memset(out, 0, sizeof *out);
  out->field_id = f->id;
  out->type = f->type;
  out->is_static = f->source == SRC_STATIC;
  out->prom_type = f->prom_type;
  out->prom_class = f->prom_class;
  out->tag = f->tag;
  out->unit = f->unit;
  out->prom_name = f->prom_name;
  out->prom_help = f->prom_help;
  return GM_ST_OK;
}

gm_status_t gm_field_by_tag(const char* tag, uint16_t* field_id) {
  if (!tag || !field_id) return GM_ST_BADPARAM;
  const FieldDef* f = field_by_tag(tag);
  if (!f) return GM_ST_NOT_FOUND;
  *field_id = f->id;
  return GM_ST_OK;
}

gm_status_t gm_field_list(uint16_t* ids, int* n) {
  if (!n) return GM_ST_BADPARAM;
  int cap = *n, cnt = field_count();
  *n = cnt;
  if (!ids || cap < cnt) return GM_ST_INSUFFICIENT_SIZE;
  for (int i = 0; i < cnt; i++) ids[i] = field_at(i).id;
  return GM_ST_OK;
}

// -------------------------------------------------------------- discovery --
gm_status_t gm_device_count(uint32_t* count) {
  REQUIRE_ENGINE();
  if (!count) return GM_ST_BADPARAM;
  return e->device_count(count);
}

static gm_status_t copy_ids(const std::vector<uint32_t>& v, uint32_t* ids, int* n) {
  int cap = *n;
  *n = (int)v.size();
  if (cap < (int)v.size() || (!ids && !v.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < v.size(); i++) ids[i] = v[i];
  return GM_ST_OK;
}

gm_status_t gm_supported_devices(uint32_t* ids, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<uint32_t> v;
  gm_status_t st = e->supported_devices(&v);
  return st != GM_ST_OK ? st : copy_ids(v, ids, n);
}

gm_status_t gm_device_attributes(uint32_t gpu, gm_device_attributes_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_ATTRIBUTES_VERSION) return GM_ST_VER_MISMATCH;
  return e->device_attributes(gpu, out);
}

gm_status_t gm_device_topology(uint32_t gpu, gm_topology_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_TOPOLOGY_VERSION) return GM_ST_VER_MISMATCH;
  return e->device_topology(gpu, out);
}

gm_status_t gm_xgmi_links(uint32_t gpu, gm_xgmi_link_t* out, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<gm_xgmi_link_t> v;
  gm_status_t st = e->xgmi_links(gpu, &v);
  if (st != GM_ST_OK) return st;
  int cap = *n;
  *n = (int)v.size();
  if (cap < (int)v.size() || (!out && !v.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return GM_ST_OK;
}

const char* gm_link_type_name(int32_t link, uint32_t hops) { return link_type_name(link, hops); }
const char* gm_link_type_description(int32_t link) { return link_type_description(link); }

// ------------------------------------------------------------------ groups --
gm_status_t gm_group_create(const char* name, gm_group_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  return e->group_create(name ? name : "", out);
}
gm_status_t gm_group_add_device(gm_group_t g, uint32_t gpu) {
  REQUIRE_ENGINE();
  return e->group_add(g, gpu);
}
gm_status_t gm_group_remove_device(gm_group_t g, uint32_t gpu) {
  REQUIRE_ENGINE();
  return e->group_remove(g, gpu);
}
gm_status_t gm_group_devices(gm_group_t g, uint32_t* ids, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<uint32_t> v;
  gm_status_t st = e->group_devices(g, &v);
  return st != GM_ST_OK ? st : copy_ids(v, ids, n);
}
gm_status_t gm_group_destroy(gm_group_t g) {
  REQUIRE_ENGINE();
  return e->group_destroy(g);
}
gm_status_t gm_field_group_create(const uint16_t* field_ids, int n, const char* name, gm_field_group_t* out) {
  REQUIRE_ENGINE();
  if (!field_ids || n <= 0 || !out) return GM_ST_BADPARAM;
  return e->field_group_create(std::vector<uint16_t>(field_ids, field_ids + n), name ? name : "", out);
}
gm_status_t gm_field_group_destroy(gm_field_group_t fg) {
  REQUIRE_ENGINE();
  return e->field_group_destroy(fg);
}
gm_status_t gm_watch_fields(gm_group_t g, gm_field_group_t fg, int64_t update_freq_us, double max_keep_age_s,
                            int32_t max_keep_samples) {
  REQUIRE_ENGINE();
  return e->watch_fields(g, fg, update_freq_us, max_keep_age_s, max_keep_samples);
}
gm_status_t gm_unwatch_fields(gm_group_t g, gm_field_group_t fg) {
  REQUIRE_ENGINE();
  return e->unwatch_fields(g, fg);
}
gm_status_t gm_update_all_fields(int wait) {
  REQUIRE_ENGINE();
  return e->update_all(wait != 0);
}
gm_status_t gm_latest_values(uint32_t gpu, const uint16_t* field_ids, int n, gm_value_t* out) {
  REQUIRE_ENGINE();
  if (!field_ids || n <= 0 || !out) return GM_ST_BADPARAM;
  return e->latest_values(gpu, field_ids, n, out);
}
gm_status_t gm_values_since(uint32_t gpu, uint16_t field_id, int64_t since_ts_us, gm_value_t* out, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<gm_value_t> v;
  gm_status_t st = e->values_since(gpu, field_id, since_ts_us, &v);
  if (st != GM_ST_OK) return st;
  int cap = *n;
  *n = (int)v.size();
  // The newest samples win when the buffer is short.
  size_t skip = v.size() > (size_t)cap ? v.size() - (size_t)cap : 0;
  for (size_t i = skip; i < v.size() && out; i++) out[i - skip] = v[i];
  return skip ? GM_ST_INSUFFICIENT_SIZE : GM_ST_OK;
}
gm_status_t gm_inject_value(uint32_t gpu, const gm_value_t* value, int32_t hold_ms) {
  REQUIRE_ENGINE();
  if (!value) return GM_ST_BADPARAM;
  return e->inject_value(gpu, *value, hold_ms);
}
gm_status_t gm_fake_ctl(const char* command) {
  REQUIRE_ENGINE();
  if (!command) return GM_ST_BADPARAM;
  return e->fake_ctl(command);
}

// ------------------------------------------------------------------ status --
gm_status_t gm_device_status(uint32_t gpu, gm_device_status_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_STATUS_VERSION) return GM_ST_VER_MISMATCH;
  return e->device_status(gpu, out);
}
gm_status_t gm_device_processes(uint32_t gpu, gm_process_t* out, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<gm_process_t> v;
  gm_status_t st = e->device_processes(gpu, &v);
  if (st != GM_ST_OK) return st;
  int cap = *n;
  *n = (int)v.size();
  if (cap < (int)v.size() || (!out && !v.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return GM_ST_OK;
}
gm_status_t gm_process_source(uint32_t gpu, char* buf, size_t cap) {
  REQUIRE_ENGINE();
  if (!buf || !cap) return GM_ST_BADPARAM;
  std::string s;
  gm_status_t st = e->process_source(gpu, &s);
  snprintf(buf, cap, "%s", s.c_str());
  return st;
}
gm_status_t gm_proc_scan(const char* proc_root, const char* pci_bdf, int render_minor, gm_process_t* out, int* n, char* source, size_t cap) {
  if (!proc_root || !n || *n < 0 || (!out && *n > 0)) return GM_ST_BADPARAM;
  gpumon::ProcScan scan;
  gpumon::scan_gpu_processes(proc_root, pci_bdf ? pci_bdf : "", render_minor, &scan);
  if (source && cap) snprintf(source, cap, "%s", gpumon::proc_source_name(scan.source));
  const int capn = *n;
  *n = (int)scan.procs.size();
  for (int i = 0; i < *n && i < capn; i++) {
    gm_process_t& o = out[i];
    memset(&o, 0, sizeof o);
    o.pid = scan.procs[i].pid;
    o.vram_bytes = scan.procs[i].vram_bytes;
    o.gtt_bytes = scan.procs[i].gtt_bytes;
    o.gfx_engine_ns = scan.procs[i].gfx_engine_ns;
    o.cu_occupancy = scan.procs[i].cu_occupancy;
    gpumon::copy_str(o.name, scan.procs[i].name);
  }
  return *n > capn ? GM_ST_INSUFFICIENT_SIZE : GM_ST_OK;
}
gm_status_t gm_metrics_table_decode(const void* raw, size_t n, char* json, size_t cap) {
  if (!raw || !json || !cap) return GM_ST_BADPARAM;
  gpumon::MetricsTable m;
  std::string why;
  if (!gpumon::metrics_table_decode(static_cast<const uint8_t*>(raw), n, &m, &why)) {
    snprintf(json, cap, "%s", why.c_str());
    return GM_ST_NOT_SUPPORTED;
  }
  const std::string j = gpumon::metrics_table_json(m);
  if (j.size() + 1 > cap) return GM_ST_INSUFFICIENT_SIZE;
  memcpy(json, j.c_str(), j.size() + 1);
  return GM_ST_OK;
}
gm_status_t gm_metrics_table_source(uint32_t gpu, char* buf, size_t cap) {
  REQUIRE_ENGINE();
  if (!buf || !cap) return GM_ST_BADPARAM;
  std::string s;
  gm_status_t st = e->metrics_table_source(gpu, &s);
  if (st == GM_ST_OK) snprintf(buf, cap, "%s", s.c_str());
  return st;
}
gm_status_t gm_device_mode(uint32_t gpu, gm_device_mode_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_DEVICE_MODE_VERSION) return GM_ST_VER_MISMATCH;
  return e->device_mode(gpu, out);
}

// ------------------------------------------------------------------ health --
gm_status_t gm_health_set(gm_group_t g, uint32_t systems) {
  REQUIRE_ENGINE();
  return e->health_set(g, systems);
}
gm_status_t gm_health_get(gm_group_t g, uint32_t* systems) {
  REQUIRE_ENGINE();
  if (!systems) return GM_ST_BADPARAM;
  return e->health_get(g, systems);
}
gm_status_t gm_health_check(gm_group_t g, gm_health_response_t* out, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  if (out && *n > 0 && out[0].version != GM_HEALTH_RESPONSE_VERSION) return GM_ST_VER_MISMATCH;
  std::vector<gm_health_response_t> v;
  gm_status_t st = e->health_check(g, &v);
  if (st != GM_ST_OK) return st;
  int cap = *n;
  *n = (int)v.size();
  if (cap < (int)v.size() || (!out && !v.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return GM_ST_OK;
}
gm_status_t gm_health_thresholds_set(const gm_health_thresholds_t* t) {
  REQUIRE_ENGINE();
  if (!t) return GM_ST_BADPARAM;
  return e->health_thresholds_set(*t);
}
gm_status_t gm_health_thresholds_get(gm_health_thresholds_t* t) {
  REQUIRE_ENGINE();
  if (!t) return GM_ST_BADPARAM;
  return e->health_thresholds_get(t);
}
gm_status_t gm_diag_result_submit(const gm_diag_result_t* r) {
  REQUIRE_ENGINE();
  if (!r) return GM_ST_BADPARAM;
  return e->diag_submit(*r);
}
gm_status_t gm_diag_results(uint32_t gpu, gm_diag_result_t* out, int* n) {
  REQUIRE_ENGINE();
  if (!n) return GM_ST_BADPARAM;
  std::vector<gm_diag_result_t> v;
  gm_status_t st = e->diag_results(gpu, &v);
  if (st != GM_ST_OK) return st;
  int cap = *n;
  *n = (int)v.size();
  if (cap < (int)v.size() || (!out && !v.empty())) return GM_ST_INSUFFICIENT_SIZE;
  for (size_t i = 0; i < v.size(); i++) out[i] = v[i];
  return GM_ST_OK;
}
const char* gm_health_system_name(uint32_t system) { return health_system_name(system); }
const char* gm_health_result_name(int32_t health) { return health_result_name(health); }

// ---------------------------------------------------------------- policies --
gm_status_t gm_policy_set(gm_group_t g, const gm_policy_t* p) {
  REQUIRE_ENGINE();
  if (!p) return GM_ST_BADPARAM;
  return e->policy_set(g, *p);
}
gm_status_t gm_policy_get(gm_group_t g, gm_policy_t* p) {
  REQUIRE_ENGINE();
  if (!p) return GM_ST_BADPARAM;
  return e->policy_get(g, p);
}
gm_status_t gm_policy_register(gm_group_t g, uint32_t conditions, gm_violation_cb cb, void* user,
                               gm_policy_reg_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  auto q = std::make_shared<PolicyQueue>();
  q->cb = cb;
  q->user = user;
  gm_status_t st = e->policy_register(
      g, conditions,
      [q](const gm_violation_t& v) {
        {
          std::lock_guard<std::mutex> l(q->mu);
          if (q->q.size() >= 1024) q->q.pop_front();
          q->q.push_back(v);
        }
        q->cv.notify_all();
        if (q->cb) q->cb(&v, q->user);
      },
      out);
  if (st != GM_ST_OK) return st;
  std::lock_guard<std::mutex> l(G().obj_mu);
  G().policy_queues[*out] = q;
  return GM_ST_OK;
}
gm_status_t gm_policy_unregister(gm_policy_reg_t reg) {
  REQUIRE_ENGINE();
  gm_status_t st = e->policy_unregister(reg);
  std::lock_guard<std::mutex> l(G().obj_mu);
  auto it = G().policy_queues.find(reg);
  if (it != G().policy_queues.end()) {
    it->second->cv.notify_all();
    G().policy_queues.erase(it);
  }
  return st;
}
gm_status_t gm_policy_wait(gm_policy_reg_t reg, int32_t timeout_ms, gm_violation_t* out) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  if (!out) return GM_ST_BADPARAM;
  std::shared_ptr<PolicyQueue> q;
  {
    std::lock_guard<std::mutex> l(G().obj_mu);
    auto it = G().policy_queues.find(reg);
    if (it == G().policy_queues.end()) return GM_ST_NOT_FOUND;
    q = it->second;
  }
  std::unique_lock<std::mutex> l(q->mu);
  if (!cv_wait_for(q->cv, l, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms), [&] { return !q->q.empty(); }))
    return GM_ST_TIMEOUT;
  *out = q->q.front();
  q->q.pop_front();
  return GM_ST_OK;
}
const char* gm_policy_condition_name(uint32_t condition) { return policy_condition_name(condition); }

// -------------------------------------------------------------- accounting --
gm_status_t gm_watch_pid_fields(gm_group_t g, int64_t update_freq_us, double max_keep_age_s,
                                int32_t max_keep_samples) {
  REQUIRE_ENGINE();
  return e->watch_pid_fields(g, update_freq_us, max_keep_age_s, max_keep_samples);
}
gm_status_t gm_pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_PID_INFO_VERSION) return GM_ST_VER_MISMATCH;
  return e->pid_info(g, pid, out);
}

// ----------------------------------------------------------- introspection --
gm_status_t gm_introspect_toggle(int enabled) {
  REQUIRE_ENGINE();
  return e->introspect_toggle(enabled != 0);
}
gm_status_t gm_introspect(gm_introspect_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  if (out->version != GM_INTROSPECT_VERSION) return GM_ST_VER_MISMATCH;
  return e->introspect(out);
}

// ------------------------------------------------------------------ events --
gm_status_t gm_event_set_create(gm_event_set_t* out) {
  REQUIRE_ENGINE();
  if (!out) return GM_ST_BADPARAM;
  auto es = std::make_shared<EventSet>();
  memset(es->mask, 0, sizeof es->mask);
  std::weak_ptr<EventSet> wes = es;
  gm_status_t st = e->event_subscribe(
      [wes](const gm_event_t& ev) {
        auto s = wes.lock();
        if (!s) return;
        std::lock_guard<std::mutex> l(s->mu);
        if (!s->registered || ev.gpu >= GM_MAX_GPUS) return;
        uint64_t m = s->mask[ev.gpu] | s->mask[GM_MAX_GPUS];
        if (!(m & (1ull << (ev.code & 63)))) return;
        if (s->q.size() >= 1024) s->q.pop_front();
        s->q.push_back(ev);
        s->cv.notify_all();
      },
      &es->sub);
  if (st != GM_ST_OK) return st;
  std::lock_guard<std::mutex> l(G().obj_mu);
  *out = G().next_es++;
  G().event_sets[*out] = es;
  return GM_ST_OK;
}

gm_status_t gm_event_register(gm_event_set_t es, uint32_t gpu, uint64_t event_mask) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  if (gpu > GM_MAX_GPUS) return GM_ST_BADPARAM;
  std::shared_ptr<EventSet> s;
  {
    std::lock_guard<std::mutex> l(G().obj_mu);
    auto it = G().event_sets.find(es);
    if (it == G().event_sets.end()) return GM_ST_NOT_FOUND;
    s = it->second;
  }
  if (!event_mask)  // every critical event: the XidCriticalError analog
    event_mask = (1ull << GM_EVENT_VMFAULT) | (1ull << GM_EVENT_GPU_PRE_RESET) | (1ull << GM_EVENT_GPU_POST_RESET) |
                 (1ull << GM_EVENT_RAS_UE) | (1ull << GM_EVENT_RAS_POISON) | (1ull << GM_EVENT_RING_HANG) |
                 (1ull << GM_EVENT_DIAG_FAIL);
  std::lock_guard<std::mutex> l(s->mu);
  s->mask[gpu] |= event_mask;
  s->registered = true;
  return GM_ST_OK;
}

gm_status_t gm_event_wait(gm_event_set_t es, int32_t timeout_ms, gm_event_t* out) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  if (!out) return GM_ST_BADPARAM;
  std::shared_ptr<EventSet> s;
  {
    std::lock_guard<std::mutex> l(G().obj_mu);
    auto it = G().event_sets.find(es);
    if (it == G().event_sets.end()) return GM_ST_NOT_FOUND;
    s = it->second;
  }
  std::unique_lock<std::mutex> l(s->mu);
  if (!cv_wait_for(s->cv, l, std::chrono::milliseconds(timeout_ms < 0 ? 0 : timeout_ms), [&] { return !s->q.empty(); }))
    return GM_ST_TIMEOUT;
  *out = s->q.front();
  s->q.pop_front();
  return GM_ST_OK;
}

gm_status_t gm_event_set_free(gm_event_set_t es) {
  REQUIRE_ENGINE();
  std::shared_ptr<EventSet> s;
  {
    std::lock_guard<std::mutex> l(G().obj_mu);
    auto it = G().event_sets.find(es);
    if (it == G().event_sets.end()) return GM_ST_NOT_FOUND;
    s = it->second;
    G().event_sets.erase(it);
  }
  return e->event_unsubscribe(s->sub);
}

// ---------------------------------------------------------------- exporter --
gm_status_t gm_exporter_create(const gm_exporter_config_t* c, gm_exporter_t* out) {
  REQUIRE_ENGINE();
  if (!c || !out) return GM_ST_BADPARAM;
  if (c->version != GM_EXPORTER_CONFIG_VERSION) return GM_ST_VER_MISMATCH;
  ExporterConfig cfg;
  cfg.profiling = c->profiling != 0;
  cfg.amd_extras = c->amd_extras != 0;
  cfg.self_metrics = c->self_metrics != 0;
  cfg.rest_api = c->rest_api != 0;
  cfg.legacy_layout = c->legacy_layout != 0;
  cfg.interval_ms = c->interval_ms > 0 ? c->interval_ms : 1000;
  if (c->gpus) cfg.gpus = c->gpus;
  if (c->output_path) cfg.output_path = c->output_path;
  if (c->listen) cfg.listen = c->listen;
  if (c->kubelet_socket) cfg.kubelet_socket = c->kubelet_socket;
  if (c->resource_names && *c->resource_names) cfg.resource_names = c->resource_names;
  if (c->pod_map_path) cfg.pod_map_path = c->pod_map_path;
  if (c->diag_results_path) cfg.diag_results_path = c->diag_results_path;
  if (c->state_path) cfg.state_path = c->state_path;
  cfg.health = c->health != 0;
  if (c->slow_interval_ms > 0) cfg.slow_interval_ms = c->slow_interval_ms;
  if (c->pod_refresh_ms > 0) cfg.pod_refresh_ms = c->pod_refresh_ms;
  auto ex = std::make_shared<Exporter>(e, cfg);
  gm_status_t st = ex->init();
  if (st != GM_ST_OK) return st;
  std::lock_guard<std::mutex> l(G().obj_mu);
  *out = G().next_exp++;
  G().exporters[*out] = ex;
  return GM_ST_OK;
}

static std::shared_ptr<Exporter> find_exporter(gm_exporter_t id) {
  std::lock_guard<std::mutex> l(G().obj_mu);
  auto it = G().exporters.find(id);
  return it == G().exporters.end() ? nullptr : it->second;
}

gm_status_t gm_exporter_render(gm_exporter_t id, char* buf, size_t* len) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  auto ex = find_exporter(id);
  if (!ex) return GM_ST_NOT_FOUND;
  if (!len) return GM_ST_BADPARAM;
  std::string text;
  ex->render(&text);
  size_t cap = *len;
  *len = text.size() + 1;
  if (!buf || cap == 0) return GM_ST_INSUFFICIENT_SIZE;
  size_t n = text.size() < cap - 1 ? text.size() : cap - 1;
  memcpy(buf, text.data(), n);
  buf[n] = 0;
  return n == text.size() ? GM_ST_OK : GM_ST_INSUFFICIENT_SIZE;
}
gm_status_t gm_exporter_sweep(gm_exporter_t id) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  auto ex = find_exporter(id);
  return ex ? ex->sweep() : GM_ST_NOT_FOUND;
}
gm_status_t gm_exporter_start(gm_exporter_t id) {
  if (!engine()) return GM_ST_UNINITIALIZED;
  auto ex = find_exporter(id);
  return ex ? ex->start() : GM_ST_NOT_FOUND;
}
gm_status_t gm_exporter_stop(gm_exporter_t id) {
  auto ex = find_exporter(id);
  if (!ex) return GM_ST_NOT_FOUND;
  ex->stop();
  return GM_ST_OK;
}
gm_status_t gm_exporter_http_port(gm_exporter_t id, int* port) {
  auto ex = find_exporter(id);
  if (!ex || !port) return GM_ST_NOT_FOUND;
  *port = ex->http_port();
  return GM_ST_OK;
}
gm_status_t gm_exporter_destroy(gm_exporter_t id) {
  std::shared_ptr<Exporter> ex;
  {
    std::lock_guard<std::mutex> l(G().obj_mu);
    auto it = G().exporters.find(id);
    if (it == G().exporters.end()) return GM_ST_NOT_FOUND;
    ex = it->second;
    G().exporters.erase(it);
  }
  ex->stop();
  return GM_ST_OK;
}

static gm_status_t copy_out(const std::string& text, char* out, size_t* len) {
  size_t cap = *len;
  *len = text.size() + 1;
  if (!out || cap < text.size() + 1) return GM_ST_INSUFFICIENT_SIZE;
  memcpy(out, text.data(), text.size());
  out[text.size()] = 0;
  return GM_ST_OK;
}

static std::mutex g_pe_mu;
static std::map<gm_pod_exporter_t, std::shared_ptr<PodExporter>> g_pod_exporters;
static uint32_t g_next_pe = 1;

gm_status_t gm_pod_exporter_start(const gm_pod_exporter_config_t* c, gm_pod_exporter_t* out) {
  if (!c || !out) return GM_ST_BADPARAM;
  if (c->version != GM_POD_EXPORTER_CONFIG_VERSION) return GM_ST_VER_MISMATCH;
  PodExporterConfig cfg;
  if (c->source_path && *c->source_path) cfg.source_path = c->source_path;
  if (c->dest_path && *c->dest_path) cfg.dest_path = c->dest_path;
  if (c->kubelet_socket && *c->kubelet_socket) cfg.kubelet_socket = c->kubelet_socket;
  if (c->resource_names && *c->resource_names) cfg.resource_names = c->resource_names;
  if (c->index_prefix && *c->index_prefix) cfg.index_prefix = c->index_prefix;
  if (c->listen) cfg.listen = c->listen;
  if (c->timeout_ms > 0) cfg.timeout_ms = c->timeout_ms;
  if (c->liveness_ms != 0) cfg.liveness_ms = c->liveness_ms < 0 ? 0 : c->liveness_ms;
  if (c->pod_refresh_ms > 0) cfg.pod_refresh_ms = c->pod_refresh_ms;
  auto pe = std::make_shared<PodExporter>(cfg);
  std::string err;
  if (!pe->start(&err)) {
    GM_LOGE("pod exporter: %s", err.c_str());
    return GM_ST_GENERIC_ERROR;
  }
  std::lock_guard<std::mutex> l(g_pe_mu);
  *out = g_next_pe++;
  g_pod_exporters[*out] = pe;
  return GM_ST_OK;
}

gm_status_t gm_pod_exporter_status(gm_pod_exporter_t id, int* alive, int* port, uint64_t* rewrites) {
  std::lock_guard<std::mutex> l(g_pe_mu);
  auto it = g_pod_exporters.find(id);
  if (it == g_pod_exporters.end()) return GM_ST_NOT_FOUND;
  if (alive) *alive = it->second->dead() ? 0 : 1;
  if (port) *port = it->second->http_port();
  if (rewrites) *rewrites = it->second->rewrites();
  return GM_ST_OK;
}

gm_status_t gm_pod_exporter_stop(gm_pod_exporter_t id) {
  std::shared_ptr<PodExporter> pe;
  {
    std::lock_guard<std::mutex> l(g_pe_mu);
    auto it = g_pod_exporters.find(id);
    if (it == g_pod_exporters.end()) return GM_ST_NOT_FOUND;
    pe = it->second;
    g_pod_exporters.erase(it);
  }
  pe->stop();
  return GM_ST_OK;
}

gm_status_t gm_render_view(int view, uint32_t id, int json, char* out, size_t* len) {
  REQUIRE_ENGINE();
  if (!len) return GM_ST_BADPARAM;
  std::string text;
  gm_status_t st;
  switch (view) {
    case GM_VIEW_DEVICE_INFO: st = view_device_info(e, id, json != 0, &text); break;
    case GM_VIEW_DEVICE_STATUS: st = view_device_status(e, id, json != 0, &text); break;
    case GM_VIEW_HEALTH: st = view_health(e, id, json != 0, &text); break;
    case GM_VIEW_PROCESS: st = view_process_info(e, id, json != 0, &text); break;
    case GM_VIEW_HOSTENGINE: st = view_hostengine(e, json != 0, &text); break;
    case GM_VIEW_TOPOLOGY: st = view_topology_matrix(e, &text); break;
    case GM_VIEW_LINKS: st = view_links(e, id, json != 0, &text); break;
    case GM_VIEW_DIAG: st = view_diag_results(e, id, json != 0, &text); break;
    default: return GM_ST_BADPARAM;
  }
  if (st != GM_ST_OK) return st;
  return copy_out(text, out, len);
}

gm_status_t gm_format_violation(const gm_violation_t* v, char* out, size_t* len) {
  if (!v || !len) return GM_ST_BADPARAM;
  return copy_out(view_violation(*v), out, len);
}

const char* gm_event_code_name(uint32_t code) { return event_code_name(code); }

gm_status_t gm_pod_relabel(const char* text, const char* map_tsv, const char* index_prefix, char* out, size_t* len) {
  if (!text || !map_tsv || !len) return GM_ST_BADPARAM;
  PodMap m = parse_pod_map_tsv(map_tsv);
  return copy_out(relabel_text(text, m, index_prefix ? index_prefix : "amdgpu"), out, len);
}

gm_status_t gm_kubelet_pod_map(const char* socket_path, const char* resource_names, int32_t timeout_ms, char* out,
                               size_t* len) {
  if (!socket_path || !len) return GM_ST_BADPARAM;
  PodMap m;
  std::string err;
  gm_status_t st = kubelet_list_pod_resources(socket_path, resource_names && *resource_names ? resource_names : "amd.com/gpu",
                                              timeout_ms > 0 ? timeout_ms : 10000, &m, &err);
  if (st != GM_ST_OK) {
    GM_LOGW("%s", err.c_str());
    return st;
  }
  return copy_out(pod_map_to_tsv(m), out, len);
}

}  // extern "C"
