// gpumon — one CLI for what the reference ships as ten sample programs
// (bindings/go/samples/nvml/{deviceInfo,dmon,processInfo},
//  bindings/go/samples/dcgm/{deviceInfo,dmon,health,hostengineStatus,policy,processInfo,topology}).
//
//   gpumon info                      static device dump (device-layer view)
//   gpumon discovery                 engine view incl. P2P topology
//   gpumon dmon [-d ms] [-c count] [-e id,id,...]
//   gpumon ps [-c count]             GPU processes
//   gpumon topo                      link matrix + CPU affinity
//   gpumon health [-c count]
//   gpumon policy [--conditions list] [--thermal C] [--power W] [--max-pages N] [-c count]
//   gpumon pidinfo -pid PID [--wait ms]
//   gpumon introspect
//   gpumon events [-c count]         wait for critical driver events
//   gpumon fields                    list the field registry
//   gpumon inject -g GPU -f FIELD -v VALUE [--hold ms]
//
// Connection flags (all sub-commands): -connect ADDR [-socket] | --start-hostengine
//                                      --backend amdsmi|fake --backend-spec SPEC
#include <cmath>
#include <map>
#include <unistd.h>

#include "cli_common.h"

static int usage() {
  fprintf(stderr,
          "usage: gpumon <info|discovery|dmon|ps|topo|links|health|diag|policy|pidinfo|introspect|events|fields|inject> [options]\n"
          "       connection: [-connect ADDR [-socket]] [--start-hostengine] [--backend amdsmi|fake] [--backend-spec S]\n");
  return 2;
}

static std::string opt(const std::vector<std::string>& a, const char* name, const char* def) {
  for (size_t i = 0; i + 1 < a.size(); i++)
    if (a[i] == name) return a[i + 1];
  return def;
}
static bool flag(const std::vector<std::string>& a, const char* name) {
  for (auto& x : a)
    if (x == name) return true;
  return false;
}

static void sleep_ms(int ms) {
  for (int i = 0; i < ms / 50 && !g_stop; i++) usleep(50000);
  if (!g_stop && ms % 50) usleep((ms % 50) * 1000);
}

static int cmd_info() {
  uint32_t n = 0;
  CLI_CHECK(gm_device_count(&n));
  for (uint32_t i = 0; i < n; i++) {
    gm_device_attributes_t a;
    a.version = GM_DEVICE_ATTRIBUTES_VERSION;
    CLI_CHECK(gm_device_attributes(i, &a));
    gm_topology_t t;
    t.version = GM_TOPOLOGY_VERSION;
    if (gm_device_topology(i, &t) != GM_ST_OK) t.num_peers = 0;
    if (i == 0) printf("Driver Version : %5s\n", a.driver_version);
    printf("GPU %12s %u\n", ":", i);
    printf("UUID           : %s\n", a.uuid);
    printf("Model          : %s\n", *a.name ? a.name : "N/A");
    printf("Path           : %s\n", *a.device_path ? a.device_path : "N/A");
    if (std::isnan(a.power_cap_w)) printf("Power          : N/A\n"); else printf("Power          : %.0f W\n", a.power_cap_w);
    if (a.vram_total_mib < 0) printf("Memory         : N/A\n"); else printf("Memory         : %lld MiB\n", (long long)a.vram_total_mib);
    printf("GfxTarget      : %s (%u XCDs, %u CUs, %s/%s)\n", *a.gfx_target ? a.gfx_target : "N/A", a.num_xcd, a.num_cu,
           *a.compute_partition ? a.compute_partition : "N/A", *a.memory_partition ? a.memory_partition : "N/A");
    if (a.numa_node < 0) printf("CPU Affinity   : N/A\n"); else printf("CPU Affinity   : NUMA node%d\n", a.numa_node);
    printf("Bus ID         : %s\n", a.pci_bdf);
    if (a.vis_vram_total_mib < 0) printf("Visible VRAM   : N/A\n"); else printf("Visible VRAM   : %lld MiB\n", (long long)a.vis_vram_total_mib);
    if (a.pcie_bandwidth_mbs < 0) printf("Bandwidth      : N/A\n"); else printf("Bandwidth      : %lld MB/s\n", (long long)a.pcie_bandwidth_mbs);
    if (a.gfx_clock_max_mhz < 0) printf("Cores          : N/A\n"); else printf("Cores          : %lld MHz\n", (long long)a.gfx_clock_max_mhz);
    if (a.mem_clock_max_mhz < 0) printf("Memory         : N/A\n"); else printf("Memory         : %lld MHz\n", (long long)a.mem_clock_max_mhz);
    printf("P2P Available  : ");
    if (!t.num_peers) printf("None\n");
    else {
      for (uint32_t k = 0; k < t.num_peers; k++)
        printf("\n                      %s - %s", t.peers[k].pci_bdf, gm_link_type_description(t.peers[k].link));
      printf("\n");
    }
    printf("---------------------------------------------------------------------\n");
  }
  return 0;
}

static int cmd_discovery(bool json) {
  uint32_t n = 0;
  CLI_CHECK(gm_device_count(&n));
  for (uint32_t i = 0; i < n; i++) {
    gm_status_t st;
    std::string s = render_view(GM_VIEW_DEVICE_INFO, i, json, &st);
    CLI_CHECK(st);
    fputs(s.c_str(), stdout);
  }
  return 0;
}

static int cmd_dmon(const std::vector<std::string>& a) {
  int interval = atoi(opt(a, "-d", "1000").c_str());
  int count = atoi(opt(a, "-c", "0").c_str());
  std::string fields = opt(a, "-e", "");
  if (interval < 100) interval = 100;
  uint32_t ids[GM_MAX_GPUS];
  int n = GM_MAX_GPUS;
  CLI_CHECK(gm_supported_devices(ids, &n));
  if (fields.empty()) {
    printf("# gpu   pwr  temp    sm   mem   enc   dec  mclk  pclk\n# Idx     W     C     %%     %%     %%     %%   MHz   MHz\n");
    for (int it = 0; !g_stop && (count == 0 || it < count); it++) {
      if (it) sleep_ms(interval);
      for (int g = 0; g < n && !g_stop; g++) {
        gm_device_status_t d;
        d.version = GM_DEVICE_STATUS_VERSION;
        CLI_CHECK(gm_device_status(ids[g], &d));
        auto cell = [&](int bit, int64_t v) {
          if ((d.valid >> bit) & 1) printf(" %5lld", (long long)v); else printf(" %5s", "-");
        };
        printf("%5u", ids[g]);
        cell(GM_DS_POWER, (int64_t)d.power_w);
        cell(GM_DS_TEMP, d.temperature_c);
        cell(GM_DS_GFX_UTIL, d.gfx_util);
        cell(GM_DS_MEM_UTIL, d.mem_util);
        cell(GM_DS_ENC_UTIL, d.enc_util);
        cell(GM_DS_DEC_UTIL, d.dec_util);
        cell(GM_DS_MEM_CLOCK, d.mem_clock_mhz);
        cell(GM_DS_GFX_CLOCK, d.gfx_clock_mhz);
        printf("\n");
      }
      fflush(stdout);
    }
    return 0;
  }
  // Explicit field list (`dmon -e 100,101,gpu_temp`): ids or tags.
  std::vector<uint16_t> fl;
  size_t start = 0;
  while (start <= fields.size()) {
    size_t c = fields.find(',', start);
    std::string tok = fields.substr(start, c == std::string::npos ? std::string::npos : c - start);
    start = c == std::string::npos ? fields.size() + 1 : c + 1;
    if (tok.empty()) continue;
    uint16_t id = 0;
    if (isdigit((unsigned char)tok[0])) id = (uint16_t)atoi(tok.c_str());
    else if (gm_field_by_tag(tok.c_str(), &id) != GM_ST_OK) {
      fprintf(stderr, "unknown field %s\n", tok.c_str());
      return 2;
    }
    gm_field_meta_t m;
    if (gm_field_meta(id, &m) != GM_ST_OK) {
      fprintf(stderr, "unknown field %s\n", tok.c_str());
      return 2;
    }
    fl.push_back(id);
  }
  gm_group_t grp;
  gm_field_group_t fg;
  CLI_CHECK(gm_group_create("gpumon-dmon", &grp));
  for (int g = 0; g < n; g++) CLI_CHECK(gm_group_add_device(grp, ids[g]));
  CLI_CHECK(gm_field_group_create(fl.data(), (int)fl.size(), "gpumon-dmon", &fg));
  CLI_CHECK(gm_watch_fields(grp, fg, (int64_t)interval * 1000, GM_DEFAULT_MAX_KEEP_AGE_S, 0));
  CLI_CHECK(gm_update_all_fields(1));
  printf("# gpu");
  for (uint16_t id : fl) {
    gm_field_meta_t m;
    gm_field_meta(id, &m);
    printf(" %14s", m.tag);
  }
  printf("\n");
  std::vector<gm_value_t> vals(fl.size());
  for (int it = 0; !g_stop && (count == 0 || it < count); it++) {
    if (it) sleep_ms(interval);
    for (int g = 0; g < n; g++) {
      CLI_CHECK(gm_latest_values(ids[g], fl.data(), (int)fl.size(), vals.data()));
      printf("%5u", ids[g]);
      for (auto& v : vals) {
        if (v.status != GM_VS_OK) printf(" %14s", "N/A");
        else if (v.type == GM_FT_DOUBLE) printf(" %14.3f", v.v.dbl);
        else if (v.type == GM_FT_STRING) printf(" %14s", v.v.str);
        else printf(" %14lld", (long long)v.v.i64);
      }
      printf("\n");
    }
    fflush(stdout);
  }
  return 0;
}

static int cmd_ps(const std::vector<std::string>& a) {
  int count = atoi(opt(a, "-c", "0").c_str());
  uint32_t n = 0;
  CLI_CHECK(gm_device_count(&n));
  printf("# gpu   pid   type  mem  Command\n# Idx     #   C/G   MiB  name\n");
  for (uint32_t g = 0; g < n; g++) {  // where the list comes from on this host (device library table, /proc statistics, open files)
    char src[32] = "";
    if (gm_process_source(g, src, sizeof src) == GM_ST_OK) fprintf(stderr, "gpu %u: process list from %s\n", g, src);  // a note, not part of the table
  }
  for (int it = 0; !g_stop && (count == 0 || it < count); it++) {
    if (it) sleep_ms(1000);
    for (uint32_t g = 0; g < n; g++) {
      std::vector<gm_process_t> ps(GM_MAX_PROCS);
      int np = (int)ps.size();
      gm_status_t st = gm_device_processes(g, ps.data(), &np);
      if (st != GM_ST_OK && st != GM_ST_INSUFFICIENT_SIZE) np = 0;
      if (np > (int)ps.size()) np = (int)ps.size();
      if (np == 0) printf("%5u %5s %5s %5s %-5s\n", g, "-", "-", "-", "-");
      for (int i = 0; i < np; i++) {
        char mem[24] = "N/A";  // a holder of the device whose memory the kernel does not report
        if (ps[i].vram_bytes != GM_PROC_MEM_UNKNOWN) snprintf(mem, sizeof mem, "%llu", (unsigned long long)(ps[i].vram_bytes >> 20));
        printf("%5u %5u %5s %5s %-5s\n", g, ps[i].pid, ps[i].type == 0 ? "C" : (ps[i].type == 1 ? "G" : "C+G"), mem, ps[i].name);
      }
    }
    fflush(stdout);
  }
  return 0;
}

static int cmd_health(const std::vector<std::string>& a) {
  int count = atoi(opt(a, "-c", "0").c_str());
  bool json = flag(a, "--json");
  uint32_t ids[GM_MAX_GPUS];
  int n = GM_MAX_GPUS;
  CLI_CHECK(gm_supported_devices(ids, &n));
  for (int it = 0; !g_stop && (count == 0 || it < count); it++) {
    if (it) sleep_ms(1000);
    for (int g = 0; g < n; g++) {
      gm_status_t st;
      std::string s = render_view(GM_VIEW_HEALTH, ids[g], json, &st);
      CLI_CHECK(st);
      fputs(s.c_str(), stdout);
      if (!json) printf("\n");
    }
    fflush(stdout);
  }
  return 0;
}

static int cmd_policy(const std::vector<std::string>& a) {
  int count = atoi(opt(a, "-c", "1").c_str());
  int timeout = atoi(opt(a, "--timeout", "0").c_str());
  std::string conds = opt(a, "--conditions", "event");
  static const std::map<std::string, uint32_t> names = {
      {"ecc", GM_POLICY_COND_ECC_UE}, {"dbe", GM_POLICY_COND_ECC_UE}, {"pcie", GM_POLICY_COND_PCIE},
      {"pages", GM_POLICY_COND_MAX_BAD_PAGES}, {"thermal", GM_POLICY_COND_THERMAL}, {"power", GM_POLICY_COND_POWER},
      {"xgmi", GM_POLICY_COND_XGMI}, {"nvlink", GM_POLICY_COND_XGMI}, {"event", GM_POLICY_COND_GPU_EVENT},
      {"xid", GM_POLICY_COND_GPU_EVENT}, {"all", GM_POLICY_COND_ALL}};
  uint32_t mask = 0;
  size_t start = 0;
  while (start <= conds.size()) {
    size_t c = conds.find(',', start);
    std::string tok = conds.substr(start, c == std::string::npos ? std::string::npos : c - start);
    start = c == std::string::npos ? conds.size() + 1 : c + 1;
    if (tok.empty()) continue;
    auto it = names.find(tok);
    if (it == names.end()) {
      fprintf(stderr, "unknown condition %s (ecc,pcie,pages,thermal,power,xgmi,event,all)\n", tok.c_str());
      return 2;
    }
    mask |= it->second;
  }
  gm_policy_t p;
  CLI_CHECK(gm_policy_get(GM_GROUP_ALL_GPUS, &p));
  p.conditions = mask;
  p.thermal_c = (uint32_t)atoi(opt(a, "--thermal", std::to_string(p.thermal_c).c_str()).c_str());
  p.power_w = (uint32_t)atoi(opt(a, "--power", std::to_string(p.power_w).c_str()).c_str());
  p.max_bad_pages = (uint32_t)atoi(opt(a, "--max-pages", std::to_string(p.max_bad_pages).c_str()).c_str());
  CLI_CHECK(gm_policy_set(GM_GROUP_ALL_GPUS, &p));
  gm_policy_reg_t reg;
  CLI_CHECK(gm_policy_register(GM_GROUP_ALL_GPUS, mask, nullptr, nullptr, &reg));
  fprintf(stderr, "Policy successfully set. Waiting for violations...\n");
  int seen = 0, waited = 0;
  while (!g_stop && (count == 0 || seen < count)) {
    gm_violation_t v;
    gm_status_t st = gm_policy_wait(reg, 250, &v);
    if (st == GM_ST_TIMEOUT) {
      waited += 250;
      if (timeout > 0 && waited >= timeout) break;
      continue;
    }
    CLI_CHECK(st);
    char buf[512];
    size_t len = sizeof buf;
    gm_format_violation(&v, buf, &len);
    printf("%s\n", buf);
    fflush(stdout);
    seen++;
  }
  gm_policy_unregister(reg);
  return 0;
}

static int cmd_pidinfo(const std::vector<std::string>& a) {
  std::string pid = opt(a, "-pid", opt(a, "--pid", "").c_str());
  if (pid.empty()) {
    fprintf(stderr, "usage: gpumon pidinfo -pid PID [--wait ms] [--json]\n");
    return 2;
  }
  int wait = atoi(opt(a, "--wait", "3000").c_str());  // warm-up of the reference sample: 3 s
  CLI_CHECK(gm_watch_pid_fields(GM_GROUP_ALL_GPUS, GM_DEFAULT_UPDATE_FREQ_US, GM_DEFAULT_MAX_KEEP_AGE_S, 0));
  fprintf(stderr, "Enabling process watches...\n");
  sleep_ms(wait);
  gm_status_t st;
  std::string s = render_view(GM_VIEW_PROCESS, (uint32_t)strtoul(pid.c_str(), nullptr, 10), flag(a, "--json"), &st);
  if (st == GM_ST_NO_DATA) {
    fprintf(stderr, "no GPU activity recorded for pid %s\n", pid.c_str());
    return 1;
  }
  CLI_CHECK(st);
  fputs(s.c_str(), stdout);
  return 0;
}

static int cmd_events(const std::vector<std::string>& a) {
  int count = atoi(opt(a, "-c", "0").c_str());
  gm_event_set_t es;
  CLI_CHECK(gm_event_set_create(&es));
  CLI_CHECK(gm_event_register(es, GM_MAX_GPUS, 0));
  fprintf(stderr, "Waiting for critical GPU events...\n");
  int seen = 0;
  while (!g_stop && (count == 0 || seen < count)) {
    gm_event_t e;
    gm_status_t st = gm_event_wait(es, 250, &e);
    if (st == GM_ST_TIMEOUT) continue;
    CLI_CHECK(st);
    printf("GPU %u (%s): event %u (%s): %s\n", e.gpu, e.uuid, e.code, gm_event_code_name(e.code), e.message);
    fflush(stdout);
    seen++;
  }
  gm_event_set_free(es);
  return 0;
}

static int cmd_fields() {
  uint16_t ids[512];
  int n = 512;
  gm_status_t st = gm_field_list(ids, &n);
  if (st != GM_ST_OK) return 1;
  printf("%-6s %-26s %-7s %-6s %s\n", "id", "tag", "type", "unit", "exported as");
  for (int i = 0; i < n; i++) {
    gm_field_meta_t m;
    gm_field_meta(ids[i], &m);
    printf("%-6u %-26s %-7s %-6s %s\n", m.field_id, m.tag, m.type == GM_FT_INT64 ? "int64" : (m.type == GM_FT_DOUBLE ? "double" : "string"),
           m.unit, m.prom_name ? m.prom_name : "-");
  }
  return 0;
}

static int cmd_inject(const std::vector<std::string>& a) {
  std::string g = opt(a, "-g", ""), f = opt(a, "-f", ""), v = opt(a, "-v", "");
  if (g.empty() || f.empty() || v.empty()) {
    fprintf(stderr, "usage: gpumon inject -g GPU -f FIELD -v VALUE [--hold ms]\n");
    return 2;
  }
  uint16_t id = 0;
  if (isdigit((unsigned char)f[0])) id = (uint16_t)atoi(f.c_str());
  else CLI_CHECK(gm_field_by_tag(f.c_str(), &id));
  gm_field_meta_t m;
  CLI_CHECK(gm_field_meta(id, &m));
  gm_value_t val;
  memset(&val, 0, sizeof val);
  val.field_id = id;
  val.type = m.type;
  val.status = GM_VS_OK;
  if (m.type == GM_FT_DOUBLE) val.v.dbl = atof(v.c_str()); else val.v.i64 = atoll(v.c_str());
  CLI_CHECK(gm_inject_value((uint32_t)atoi(g.c_str()), &val, atoi(opt(a, "--hold", "0").c_str())));
  printf("Successfully injected field %u (%s) on GPU %s.\n", id, m.tag, g.c_str());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) return usage();
  std::string cmd = argv[1];
  std::vector<std::string> args(argv + 2, argv + argc);
  CliConn conn;
  if (!parse_conn(args, &conn)) return usage();
  if (cmd == "fields") return cmd_fields();
  if (cmd == "-h" || cmd == "--help" || cmd == "help") return usage();
  install_signals();
  gm_status_t st = cli_init(conn);
  if (st != GM_ST_OK) {
    fprintf(stderr, "error: cannot initialise the engine: %s\n", gm_strerror(st));
    return 1;
  }
  int rc;
  if (cmd == "info") rc = cmd_info();
  else if (cmd == "discovery") rc = cmd_discovery(flag(args, "--json"));
  else if (cmd == "dmon") rc = cmd_dmon(args);
  else if (cmd == "ps") rc = cmd_ps(args);
  else if (cmd == "topo" || cmd == "topology") {
    std::string s = render_view(GM_VIEW_TOPOLOGY, 0, false, &st);
    if (st == GM_ST_OK) fputs(s.c_str(), stdout); else fprintf(stderr, "error: %s\n", gm_strerror(st));
    rc = st == GM_ST_OK ? 0 : 1;
  } else if (cmd == "links" || cmd == "diag") {
    // gpumon links|diag [-g GPU] [--json]: every GPU unless one is named
    uint32_t n = 0;
    gm_device_count(&n);
    const std::string g = opt(args, "-g", "");
    rc = 0;
    for (uint32_t i = 0; i < n; i++) {
      if (!g.empty() && (uint32_t)atoi(g.c_str()) != i) continue;
      std::string s = render_view(cmd == "links" ? GM_VIEW_LINKS : GM_VIEW_DIAG, i, flag(args, "--json"), &st);
      if (st == GM_ST_OK) fputs(s.c_str(), stdout);
      else {
        fprintf(stderr, "GPU %u: %s\n", i, gm_strerror(st));
        if (st != GM_ST_NOT_SUPPORTED) rc = 1;
      }
    }
  } else if (cmd == "health") rc = cmd_health(args);
  else if (cmd == "policy") rc = cmd_policy(args);
  else if (cmd == "pidinfo") rc = cmd_pidinfo(args);
  else if (cmd == "introspect" || cmd == "status") {
    gm_introspect_toggle(1);
    sleep_ms(atoi(opt(args, "--wait", "1000").c_str()));
    std::string s = render_view(GM_VIEW_HOSTENGINE, 0, flag(args, "--json"), &st);
    if (st == GM_ST_OK) fputs(s.c_str(), stdout); else fprintf(stderr, "error: %s\n", gm_strerror(st));
    rc = st == GM_ST_OK ? 0 : 1;
  } else if (cmd == "events") rc = cmd_events(args);
  else if (cmd == "inject") rc = cmd_inject(args);
  else rc = usage();
  gm_shutdown();
  return rc;
}
