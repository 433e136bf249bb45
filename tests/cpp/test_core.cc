// C++ tests of the core library, meant to run under ThreadSanitizer and
// AddressSanitizer as well as plainly (make test / make tsan / make asan):
// the sampler <-> reader hand-off, concurrent control-plane calls, the RPC
// server with several clients, and the text parsers on hostile input.
// Plain asserts; no test framework.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <cassert>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <thread>

#include "core/common/json.h"
#include "core/common/util.h"
#include "core/exporter/exporter.h"
#include "core/fields.h"
#include "core/prof/prof_child.h"
#include "core/rpc/server.h"
#include "gpumon.h"

using namespace gpumon;

#define CHECK(x)                                                        \
  do {                                                                  \
    if (!(x)) {                                                         \
      fprintf(stderr, "%s:%d: CHECK failed: %s\n", __FILE__, __LINE__, #x); \
      abort();                                                          \
    }                                                                   \
  } while (0)
#define OK(expr) CHECK((expr) == GM_ST_OK)

static void init_fake(const char* spec) {
  gm_init_params_t p;
  memset(&p, 0, sizeof p);
  p.version = GM_INIT_PARAMS_VERSION;
  p.mode = GM_MODE_EMBEDDED;
  p.backend = "fake";
  p.backend_spec = spec;
  OK(gm_init(&p));
}

// -- field registry invariants -------------------------------------------------
static void test_registry() {
  std::vector<uint16_t> ids(1024);
  int n = (int)ids.size();
  OK(gm_field_list(ids.data(), &n));
  CHECK(n > 60 && n == field_count());
  for (int i = 0; i < n; i++) {
    gm_field_meta_t m;
    OK(gm_field_meta(ids[i], &m));
    CHECK(m.field_id == ids[i] && m.tag && m.tag[0]);
    uint16_t back = 0;
    OK(gm_field_by_tag(m.tag, &back));
    CHECK(back == ids[i]);
    CHECK(field_slot(ids[i]) == i);
    for (int j = 0; j < i; j++) CHECK(ids[j] != ids[i]);  // ids are unique
  }
  CHECK(gm_field_meta(9999, nullptr) != GM_ST_OK);
  printf("registry: %d fields ok\n", n);
}

// -- sampler vs readers vs control plane, all at once ---------------------------
static void test_concurrent_engine() {
  init_fake("gpus=8,seed=3");
  uint32_t ngpu = 0;
  OK(gm_device_count(&ngpu));
  CHECK(ngpu == 8);
  std::vector<uint16_t> all(1024);
  int nf = (int)all.size();
  OK(gm_field_list(all.data(), &nf));
  gm_group_t g;
  gm_field_group_t fg;
  OK(gm_group_create("stress", &g));
  for (uint32_t i = 0; i < ngpu; i++) OK(gm_group_add_device(g, i));
  OK(gm_field_group_create(all.data(), nf, "stress", &fg));
  OK(gm_watch_fields(g, fg, 1000, 1.0, 0));  // 1 kHz: the sampler is always mid-publish
  OK(gm_health_set(g, GM_HEALTH_WATCH_ALL));
  gm_policy_t pol;
  memset(&pol, 0, sizeof pol);
  pol.version = GM_POLICY_VERSION;
  pol.conditions = GM_POLICY_COND_ALL;
  pol.max_bad_pages = 10;
  pol.thermal_c = 60;
  pol.power_w = 900;
  pol.hysteresis_pct = 3;
  OK(gm_policy_set(g, &pol));

  std::atomic<bool> stop{false};
  std::atomic<uint64_t> reads{0}, violations{0}, torn{0};
  std::vector<std::thread> th;
  for (int r = 0; r < 4; r++)
    th.emplace_back([&, r] {
      uint16_t ids[] = {GM_FI_GPU_TEMP, GM_FI_POWER_USAGE, GM_FI_GFX_CLOCK, GM_FI_VRAM_USED, GM_FI_VRAM_FREE, GM_FI_VRAM_TOTAL};
      gm_value_t v[6];
      while (!stop) {
        uint32_t gpu = (uint32_t)(reads.load() + r) % 8;
        OK(gm_latest_values(gpu, ids, 6, v));
        // One call = one published state: used + free must add up to total whenever all three are there.
        if (v[3].status == GM_VS_OK && v[4].status == GM_VS_OK && v[5].status == GM_VS_OK &&
            v[3].v.i64 + v[4].v.i64 != v[5].v.i64)
          torn++;
        gm_device_status_t ds;
        memset(&ds, 0, sizeof ds);
        ds.version = GM_DEVICE_STATUS_VERSION;
        OK(gm_device_status(gpu, &ds));
        std::vector<gm_value_t> hist(256);
        int hn = (int)hist.size();
        gm_status_t st = gm_values_since(gpu, GM_FI_GPU_TEMP, 0, hist.data(), &hn);
        CHECK(st == GM_ST_OK || st == GM_ST_INSUFFICIENT_SIZE);
        for (int i = 1; i < hn && i < 256; i++) CHECK(hist[i].ts_us >= hist[i - 1].ts_us);
        reads++;
      }
    });
  th.emplace_back([&] {  // exporter rendering from the same cache
    gm_exporter_config_t c;
    memset(&c, 0, sizeof c);
    c.version = GM_EXPORTER_CONFIG_VERSION;
    c.interval_ms = 100;
    c.profiling = c.amd_extras = c.self_metrics = 1;
    c.health = 1;  // health verdicts per sweep and violation counts from the notifier thread, next to the renders
    // a diagnostics report that is rewritten while it is being exported, and a state file
    char report[64], state[64];
    snprintf(report, sizeof report, "/tmp/gpumon-test-%d-diag.json", (int)getpid());
    snprintf(state, sizeof state, "/tmp/gpumon-test-%d-state.json", (int)getpid());
    auto write_report = [&](int errors) {
      std::string tmp = std::string(report) + ".tmp";
      FILE* f = fopen(tmp.c_str(), "w");
      CHECK(f);
      fprintf(f, "{\"gpu\":3,\"timestamp\":1700000000,\"memtest\":{\"bytes_tested\":1024,\"errors\":%d},\"result\":\"%s\"}\n", errors,
              errors ? "fail" : "pass");
      fclose(f);
      CHECK(rename(tmp.c_str(), report) == 0);
    };
    write_report(0);
    c.diag_results_path = report;
    c.state_path = state;
    gm_exporter_t ex;
    OK(gm_exporter_create(&c, &ex));
    std::vector<char> buf(1 << 20);
    int round = 0;
    OK(gm_exporter_sweep(ex));
    while (!stop) {
      size_t len = buf.size();
      if (round % 10 == 0) {
        gm_status_t sw = gm_exporter_sweep(ex);  // the injector fails whole GPUs now and then
        CHECK(sw == GM_ST_OK || sw == GM_ST_GENERIC_ERROR || sw == GM_ST_STALE_DATA);
      }
      OK(gm_exporter_render(ex, buf.data(), &len));
      CHECK(strstr(buf.data(), "gpumon_health_overall{gpu=\"0\"") && strstr(buf.data(), "gpumon_policy_violations_total{gpu=\"7\""));
      CHECK(len > 1000 && strstr(buf.data(), "# TYPE dcgm_gpu_temp gauge"));  // single GPUs may be blanked by the injector
      CHECK(strstr(buf.data(), "gpumon_diag_memtest_errors{gpu=\"3\"}"));
      if (++round % 50 == 0) write_report(round % 100 == 0);
    }
    OK(gm_exporter_destroy(ex));  // writes the state file
    FILE* f = fopen(state, "r");
    CHECK(f);
    if (f) fclose(f);
    unlink(report);
    unlink(state);
  });
  th.emplace_back([&] {  // fault injector
    std::mt19937 rng(7);
    const char* cmds[] = {"set gpu=%u field=150 value=95", "clear gpu=%u field=150", "ecc gpu=%u block=0 ue=1",
                          "link_down gpu=%u link=2", "link_up gpu=%u link=2", "event gpu=%u code=5 msg=x",
                          "unsupported gpu=%u field=155", "pcie_replay gpu=%u count=3", "fail gpu=%u", "fail gpu=%u sources=0"};
    while (!stop) {
      char b[128];
      snprintf(b, sizeof b, cmds[rng() % 10], (unsigned)(rng() % 8));
      gm_fake_ctl(b);
      usleep(500);
    }
  });
  th.emplace_back([&] {  // registrations come and go while violations are being delivered
    while (!stop) {
      gm_policy_reg_t reg;
      OK(gm_policy_register(g, GM_POLICY_COND_ALL, [](const gm_violation_t*, void* u) -> int { (*(std::atomic<uint64_t>*)u)++; return 0; },
                            &violations, &reg));
      usleep(3000);
      gm_violation_t v;
      gm_policy_wait(reg, 1, &v);
      OK(gm_policy_unregister(reg));
    }
  });
  th.emplace_back([&] {  // group / watch churn next to the running sampler
    int k = 0;
    while (!stop) {
      gm_group_t g2;
      gm_field_group_t f2;
      char name[32];
      snprintf(name, sizeof name, "churn-%d", k++);
      OK(gm_group_create(name, &g2));
      OK(gm_group_add_device(g2, k % 8));
      uint16_t ids[] = {GM_FI_GPU_TEMP, GM_FI_ECC_UE_VOLATILE};
      OK(gm_field_group_create(ids, 2, name, &f2));
      OK(gm_watch_fields(g2, f2, 2000, 0.5, 16));
      gm_health_response_t hr[GM_MAX_GPUS];
      int hn = GM_MAX_GPUS;
      hr[0].version = GM_HEALTH_RESPONSE_VERSION;
      OK(gm_health_set(g2, GM_HEALTH_WATCH_ALL));
      OK(gm_health_check(g2, hr, &hn));
      OK(gm_unwatch_fields(g2, f2));
      OK(gm_field_group_destroy(f2));
      OK(gm_group_destroy(g2));
    }
  });
  th.emplace_back([&] {  // diagnostics verdicts, link tables and health levels next to health checks and the sampler
    int k = 0;
    while (!stop) {
      gm_diag_result_t d;
      memset(&d, 0, sizeof d);
      d.version = GM_DIAG_RESULT_VERSION;
      d.gpu = (uint32_t)(k % 8);
      snprintf(d.test, sizeof d.test, "%s", k % 3 ? "memtest" : "xgmi_p2p");
      d.system = k % 3 ? GM_HEALTH_WATCH_MEM : GM_HEALTH_WATCH_XGMI;
      d.health = k % 5 == 0 ? GM_HEALTH_FAIL : k % 5 == 1 ? GM_HEALTH_WARN : GM_HEALTH_PASS;
      d.value = k;
      snprintf(d.message, sizeof d.message, "round %d", k);
      OK(gm_diag_result_submit(&d));
      gm_diag_result_t got[16];
      int n = 16;
      OK(gm_diag_results(d.gpu, got, &n));
      CHECK(n >= 1 && n <= 2);
      gm_xgmi_link_t links[GM_MAX_XGMI_LINKS];
      int nl = GM_MAX_XGMI_LINKS;
      OK(gm_xgmi_links(d.gpu, links, &nl));
      CHECK(nl == 7);
      for (int i = 0; i < nl; i++) CHECK(links[i].peer_gpu >= 0 && links[i].peer_gpu < 8 && (uint32_t)links[i].peer_gpu != d.gpu);
      gm_health_thresholds_t t;
      memset(&t, 0, sizeof t);
      t.version = GM_HEALTH_THRESHOLDS_VERSION;
      OK(gm_health_thresholds_get(&t));
      t.hotspot_warn_c = 90 + k % 10;
      OK(gm_health_thresholds_set(&t));
      k++;
    }
  });
  sleep(2);
  stop = true;
  for (auto& t : th) t.join();
  gm_introspect_t in;
  memset(&in, 0, sizeof in);
  in.version = GM_INTROSPECT_VERSION;
  OK(gm_introspect(&in));
  printf("concurrent engine: %llu reads, %llu violations, %llu sample cycles, torn=%llu\n", (unsigned long long)reads.load(),
         (unsigned long long)violations.load(), (unsigned long long)in.sample_cycles, (unsigned long long)torn.load());
  CHECK(reads > 100 && in.sample_cycles > 100 && torn == 0);
  OK(gm_shutdown());
}

// -- RPC: several clients against one server -----------------------------------
static void test_rpc() {
  init_fake("gpus=4,seed=1");
  char path[] = "/tmp/gpumon-test-XXXXXX";
  int fd = mkstemp(path);
  close(fd);
  unlink(path);
  OK(gm_server_start(path, 1));
  std::atomic<uint64_t> got{0};
  std::vector<std::thread> th;
  for (int c = 0; c < 4; c++)
    th.emplace_back([&, c] {
      gm_status_t st;
      std::string err;
      auto eng = make_remote_engine(path, true, 3000, &st, &err);
      CHECK(eng && st == GM_ST_OK);
      uint32_t n = 0;
      OK(eng->device_count(&n));
      CHECK(n == 4);
      gm_group_t g;
      gm_field_group_t fg;
      OK(eng->group_create("c" + std::to_string(c), &g));
      OK(eng->group_add(g, c));
      OK(eng->field_group_create({GM_FI_GPU_TEMP, GM_FI_POWER_USAGE}, "c" + std::to_string(c), &fg));
      OK(eng->watch_fields(g, fg, 2000, 1.0, 0));
      gm_policy_t pol;
      memset(&pol, 0, sizeof pol);
      pol.version = GM_POLICY_VERSION;
      pol.conditions = GM_POLICY_COND_THERMAL;
      pol.thermal_c = 80;
      pol.hysteresis_pct = 3;
      OK(eng->policy_set(g, pol));
      gm_policy_reg_t reg;
      OK(eng->policy_register(g, GM_POLICY_COND_THERMAL, [&](const gm_violation_t& v) { if (v.gpu == (uint32_t)c) got++; }, &reg));
      OK(eng->fake_ctl("set gpu=" + std::to_string(c) + " field=150 value=97"));
      for (int i = 0; i < 200; i++) {
        uint16_t ids[] = {GM_FI_GPU_TEMP, GM_FI_POWER_USAGE};
        gm_value_t v[2];
        OK(eng->update_all(i % 20 == 0));
        OK(eng->latest_values(c, ids, 2, v));
        gm_device_status_t ds;
        memset(&ds, 0, sizeof ds);
        ds.version = GM_DEVICE_STATUS_VERSION;
        OK(eng->device_status(c, &ds));
      }
      if (c % 2) OK(eng->policy_unregister(reg));  // the others leave the clean-up to the disconnect
    });
  for (auto& t : th) t.join();
  CHECK(got >= 4);
  {
    // A request the engine refuses before it writes anything (GPU out of range, introspection off) must not bring
    // back bytes of the server's stack: beyond the version word the reply struct is all zero.
    gm_status_t st;
    std::string err;
    auto eng = make_remote_engine(path, true, 3000, &st, &err);
    CHECK(eng && st == GM_ST_OK);
    auto all_zero_after_version = [](const void* p, size_t n) {
      const unsigned char* b = (const unsigned char*)p;
      for (size_t i = sizeof(uint32_t); i < n; i++)
        if (b[i]) return false;
      return true;
    };
    gm_device_attributes_t a;
    memset(&a, 0xAA, sizeof a);
    a.version = GM_DEVICE_ATTRIBUTES_VERSION;
    CHECK(eng->device_attributes(9999, &a) != GM_ST_OK && all_zero_after_version(&a, sizeof a));
    gm_device_status_t ds;
    memset(&ds, 0xAA, sizeof ds);
    ds.version = GM_DEVICE_STATUS_VERSION;
    CHECK(eng->device_status(9999, &ds) != GM_ST_OK && all_zero_after_version(&ds, sizeof ds));
    gm_topology_t tp;
    memset(&tp, 0xAA, sizeof tp);
    tp.version = GM_TOPOLOGY_VERSION;
    CHECK(eng->device_topology(9999, &tp) != GM_ST_OK && all_zero_after_version(&tp, sizeof tp));
    gm_device_mode_t dm;
    memset(&dm, 0xAA, sizeof dm);
    dm.version = GM_DEVICE_MODE_VERSION;
    CHECK(eng->device_mode(9999, &dm) != GM_ST_OK && all_zero_after_version(&dm, sizeof dm));
  }
  gm_group_t g;
  OK(gm_group_create("c0", &g));  // names of disconnected clients are free again
  OK(gm_server_stop());
  OK(gm_shutdown());
  printf("rpc: %llu violations over 4 connections ok\n", (unsigned long long)got.load());
}

// -- parsers on hostile input ----------------------------------------------------
static void test_parsers() {
  Json j;
  CHECK(json_parse("{\"a\":[1,2,{\"b\":\"x\\n\\u0041\"}],\"c\":-1.5e2,\"d\":null,\"e\":true}", &j));
  CHECK(j.get("a") && j.get("a")->arr.size() == 3 && j.get("a")->arr[2].get("b")->str == "x\nA");
  CHECK(j.get("c")->num == -150.0);
  const char* bad[] = {"", "{", "{\"a\":}", "[1,,2]", "{\"a\":1}x", "\"unterminated", "{\"a\":tru}", "[1e999999]x", "nul"};
  for (const char* b : bad) CHECK(!json_parse(b, &j));
  std::mt19937 rng(11);
  const char alphabet[] = "{}[]\",:\\ntruefalsn0123456789.-eE \n\tabc\x01\xff";
  for (int i = 0; i < 20000; i++) {
    std::string s;
    int len = rng() % 64;
    for (int k = 0; k < len; k++) s += alphabet[rng() % (sizeof alphabet - 1)];
    json_parse(s, &j);  // must not crash or hang
    PodMap m;
    parse_device_plugin_checkpoint(s, "amd.com/gpu", &m);
  }
  std::string deep(100000, '[');
  CHECK(!json_parse(deep, &j));  // nesting is bounded, no stack overflow

  PodMap m = parse_pod_map_tsv("GPU-1\tpod\tns\tc\n#comment\nbroken line\namdgpu3\tp2\t\tc2\n");
  CHECK(m.size() == 2 && m["GPU-1"].pod == "pod" && m["amdgpu3"].ns.empty());
  const char la[] = "{}\"\\,= \nabcgpuuid0123#";
  for (int i = 0; i < 20000; i++) {
    std::string s;
    int len = rng() % 96;
    for (int k = 0; k < len; k++) s += la[rng() % (sizeof la - 1)];
    std::string out = relabel_text(s, m, "amdgpu");
    CHECK(out.size() >= s.size());
  }
  std::string text = "m{gpu=\"3\",uuid=\"GPU-x\"} 1\nm{gpu=\"0\",uuid=\"GPU-1\"} 2\n";
  std::string out = relabel_text(text, m, "amdgpu");
  CHECK(out.find("m{gpu=\"3\",uuid=\"GPU-x\",pod_name=\"p2\",pod_namespace=\"\",container_name=\"c2\"} 1\n") == 0);
  CHECK(out.find("uuid=\"GPU-1\",pod_name=\"pod\"") != std::string::npos);
  CHECK(relabel_text(out, m, "amdgpu") == out);
  printf("parsers ok\n");
}

// -- counter windows from a probe child ---------------------------------------------
static void test_prof_child() {
  std::string pci;
  ProfWindow w;
  CHECK(parse_probe_line("sample pci=0000:5D:00.0 window_s=0.750 gpu_active=99.1 cu_active=nan mem_pipe_active=61 mfma_active=84.25 clock_mhz=1934", &pci, &w));
  CHECK(pci == "0000:5d:00.0" && w.window_s == 0.75 && w.gpu_active == 99.1 && std::isnan(w.cu_active) && w.mfma_active == 84.25 && w.clock_mhz == 1934);
  CHECK(parse_probe_line("sample pci=0000:05:00.0 mfma_active=140 gpu_active=-3", &pci, &w) && std::isnan(w.mfma_active) && std::isnan(w.gpu_active));
  CHECK(!parse_probe_line("agents: 1", &pci, &w) && !parse_probe_line("sample window_s=1", &pci, &w) && !parse_probe_line("", &pci, &w));
  std::mt19937 rng(5);
  const char al[] = "sample pci=:.0123456789abcdefnan_ =\n\x01\xff";
  for (int i = 0; i < 20000; i++) {
    std::string s = rng() % 2 ? "sample " : "";
    int len = rng() % 80;
    for (int k = 0; k < len; k++) s += al[rng() % (sizeof al - 1)];
    parse_probe_line(s.c_str(), &pci, &w);  // must not crash
  }

  char dir[] = "/tmp/gpumon-probe-XXXXXX";
  CHECK(mkdtemp(dir));
  auto script = [&](const char* name, const char* body) {
    std::string path = std::string(dir) + "/" + name;
    FILE* f = fopen(path.c_str(), "w");
    CHECK(f);
    fprintf(f, "#!/bin/sh\n%s", body);
    fclose(f);
    chmod(path.c_str(), 0755);
    return path;
  };
  {
    ProfChildSampler s(script("ok.sh", "echo 'agents: 2'\necho 'sample pci=0000:05:00.0 window_s=0.3 gpu_active=50 cu_active=40 mem_pipe_active=30 mfma_active=20 clock_mhz=2000'\n"), 1.0, 50);
    s.start();
    for (int i = 0; i < 100 && !s.latest("0000:05:00.0", &w); i++) usleep(20000);
    CHECK(s.latest("0000:05:00.0", &w) && w.mfma_active == 20 && w.cu_active == 40);
    CHECK(s.latest("0000:05:00.0", &w) && w.taken_mono_us > 0);
    CHECK(!s.latest("0000:15:00.0", &w));  // a GPU the probe did not report
    CHECK(s.runs() >= 1 && s.failures() == 0);
    // readers from several threads while the loop keeps replacing the window
    std::atomic<bool> stop{false};
    std::vector<std::thread> readers;
    for (int t = 0; t < 3; t++)
      readers.emplace_back([&] {
        ProfWindow x;
        while (!stop) s.latest("0000:05:00.0", &x);
      });
    usleep(1300000);
    stop = true;
    for (auto& t : readers) t.join();
    CHECK(s.runs() >= 2);
  }
  {
    ProfChildSampler s(script("fail.sh", "echo 'gmp_open: no device' >&2\nexit 1\n"), 1.0, 50);
    s.start();
    for (int i = 0; i < 100 && s.runs() == 0; i++) usleep(20000);
    CHECK(s.runs() >= 1 && s.failures() >= 1 && !s.latest("0000:05:00.0", &w));
  }
  {
    // a probe that never ends: destroying the sampler ends it (the exact child) and returns promptly
    ProfChildSampler s(script("hang.sh", "exec sleep 300\n"), 1.0, 50);
    s.start();
    usleep(200000);
    const int64_t t0 = monotonic_us();
    (void)t0;
  }
  { ProfChildSampler s(std::string(dir) + "/missing", 1.0, 50); s.start(); usleep(100000); CHECK(s.failures() >= 1); }
  std::string cmd = std::string("rm -rf ") + dir;
  CHECK(system(cmd.c_str()) == 0);
  printf("prof child ok\n");
}

// -- refcount + direct layer ------------------------------------------------------
// -- the exporter's threads against each other ----------------------------------
// Sweeps at the 100 ms floor (one with a file sink, one that only keeps the latch), the pod-map thread re-reading a
// file that changes under it, the health check, and scrapers that render on demand: everything that touches the
// exporter's state at once.
static std::string http_get(int port, const char* path) {
  int fd = socket(AF_INET, SOCK_STREAM, 0);
  CHECK(fd >= 0);
  sockaddr_in sa;
  memset(&sa, 0, sizeof sa);
  sa.sin_family = AF_INET;
  sa.sin_port = htons((uint16_t)port);
  sa.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
  CHECK(connect(fd, (sockaddr*)&sa, sizeof sa) == 0);
  const std::string req = std::string("GET ") + path + " HTTP/1.0\r\n\r\n";
  CHECK(write(fd, req.data(), req.size()) == (ssize_t)req.size());
  std::string body;
  char buf[65536];
  ssize_t n;
  while ((n = read(fd, buf, sizeof buf)) > 0) body.append(buf, (size_t)n);
  close(fd);
  return body;
}

static void test_exporter_threads() {
  init_fake("gpus=4,seed=3");
  char dir[] = "/tmp/gpumon-cpp-XXXXXX";
  CHECK(mkdtemp(dir));
  const std::string out = std::string(dir) + "/dcgm.prom", map = std::string(dir) + "/pods.tsv", state = std::string(dir) + "/state.json";
  auto write_map = [&](int gen) {
    std::string text;
    for (int g = 0; g < 4; g++) text += strprintf("amdgpu%d\tpod-%d-%d\tns%d\tmain\n", g, g, gen, gen % 3);
    CHECK(write_file_atomic(map, text.data(), text.size()));
  };
  write_map(0);
  gm_exporter_t ex[2];
  for (int i = 0; i < 2; i++) {
    gm_exporter_config_t c;
    memset(&c, 0, sizeof c);
    c.version = GM_EXPORTER_CONFIG_VERSION;
    c.profiling = c.amd_extras = c.self_metrics = c.health = 1;
    c.interval_ms = 100;
    c.slow_interval_ms = 300;
    c.pod_refresh_ms = 100;
    c.listen = "127.0.0.1:0";
    c.pod_map_path = map.c_str();
    if (i == 0) c.output_path = out.c_str(), c.state_path = state.c_str();
    OK(gm_exporter_create(&c, &ex[i]));
    OK(gm_exporter_start(ex[i]));
  }
  int port[2];
  for (int i = 0; i < 2; i++) OK(gm_exporter_http_port(ex[i], &port[i]));
  std::atomic<bool> stop{false};
  std::atomic<int> scrapes{0}, with_pods{0};
  std::vector<std::thread> th;
  for (int t = 0; t < 4; t++)
    th.emplace_back([&, t] {
      while (!stop) {
        const std::string body = http_get(port[t & 1], t < 2 ? "/metrics" : "/gpu/metrics");
        CHECK(body.find("200 OK") != std::string::npos && body.find("dcgm_gpu_temp{gpu=\"3\"") != std::string::npos);
        if (body.find("pod_name=\"pod-") != std::string::npos) with_pods++;
        scrapes++;
      }
    });
  th.emplace_back([&] {  // the node changes under the exporters
    for (int gen = 1; !stop; gen++) {
      write_map(gen);
      gm_fake_ctl(gen & 1 ? "load gpu=1 util=90" : "load gpu=1 util=0");
      gm_fake_ctl(gen % 5 == 0 ? "ecc gpu=2 ue=1" : "set gpu=0 field=150 value=55");
      usleep(20000);
    }
  });
  usleep(1500000);
  stop = true;
  for (auto& t : th) t.join();
  for (int i = 0; i < 2; i++) {
    OK(gm_exporter_stop(ex[i]));
    OK(gm_exporter_destroy(ex[i]));
  }
  std::string text;
  CHECK(read_file(out, &text) && text.find("dcgm_gpu_last_not_idle_time{gpu=\"1\"") != std::string::npos);
  CHECK(scrapes > 20 && with_pods > 0);
  OK(gm_shutdown());
  printf("exporter threads: %d scrapes (%d with pod labels) against two exporters at the 100 ms floor ok\n", scrapes.load(), with_pods.load());
}

static void test_lifecycles() {
  CHECK(gm_shutdown() == GM_ST_UNINITIALIZED);
  init_fake("gpus=2");
  init_fake("gpus=5");
  CHECK(gm_init_refcount() == 2);
  OK(gm_shutdown());
  uint32_t n = 0;
  OK(gm_device_count(&n));
  CHECK(n == 2);
  OK(gm_shutdown());
  CHECK(gm_device_count(&n) == GM_ST_UNINITIALIZED);
  for (int i = 0; i < 20; i++) {  // start/stop leaves no thread or descriptor behind
    init_fake("gpus=8");
    OK(gm_update_all_fields(1));
    OK(gm_shutdown());
  }
  OK(gm_smi_init("fake", "gpus=3"));
  std::vector<std::thread> th;
  for (int t = 0; t < 4; t++)
    th.emplace_back([t] {
      for (int i = 0; i < 500; i++) {
        gm_device_status_t ds;
        memset(&ds, 0, sizeof ds);
        ds.version = GM_DEVICE_STATUS_VERSION;
        OK(gm_smi_device_status(t % 3, &ds));
        CHECK(ds.valid & (1ull << GM_DS_TEMP));
      }
    });
  for (auto& t : th) t.join();
  OK(gm_smi_shutdown());
  CHECK(gm_smi_shutdown() == GM_ST_UNINITIALIZED);
  printf("lifecycles ok\n");
}

int main() {
  gm_set_log_level(0);
  test_lifecycles();
  init_fake("gpus=1");
  test_registry();
  OK(gm_shutdown());
  test_parsers();
  test_prof_child();
  test_concurrent_engine();
  test_exporter_threads();
  test_rpc();
  printf("ALL OK\n");
  return 0;
}
