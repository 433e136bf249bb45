// Shared by the command-line tools: connection flags, init helper, signal flag.
// Connection flags follow the reference samples: `-connect host[:port]` with
// `-socket` for unix sockets (bindings/go/samples/dcgm/deviceInfo/main.go:36-39);
// `--start-hostengine` spawns a private daemon like its topology sample
// (bindings/go/samples/dcgm/topology/main.go:30).
#pragma once

#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "gpumon.h"

struct CliConn {
  std::string connect, backend, backend_spec, hostengine;
  bool is_socket = false, start_hostengine = false;
  // --health-<level> VALUE: levels of the health checks that differ from the defaults (gm_health_thresholds_t)
  std::vector<std::pair<std::string, std::string>> health;
};

static volatile sig_atomic_t g_stop = 0;
static void on_signal(int) { g_stop = 1; }
static inline void install_signals() {
  struct sigaction sa;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_signal;
  sigaction(SIGINT, &sa, nullptr);
  sigaction(SIGTERM, &sa, nullptr);
  sigaction(SIGHUP, &sa, nullptr);
  sigaction(SIGQUIT, &sa, nullptr);
  signal(SIGPIPE, SIG_IGN);
}

// One configuration for flags, environment and file.  `--config FILE` is replaced, in place, by the flags
// written in FILE (whitespace-separated words, one or more per line; `#` starts a comment; a word may be
// quoted with "..."), and the words of $GPUMON_ARGS are put in front of the command line, so a flag given
// on the command line wins over the file, which wins over the environment.  Returns false when a file
// cannot be read.  The reference has only flags and compile-time constants (watchers.go:15-21).
static inline bool split_words(const std::string& text, std::vector<std::string>* out) {
  std::string cur;
  bool in_word = false, quoted = false;
  for (size_t i = 0; i <= text.size(); i++) {
    const char ch = i < text.size() ? text[i] : '\n';
    if (quoted) {
      if (ch == '"') quoted = false;
      else cur += ch;
      continue;
    }
    if (ch == '#') {  // comment until the end of the line
      while (i < text.size() && text[i] != '\n') i++;
    }
    if (ch == '"') { quoted = in_word = true; continue; }
    if (ch == ' ' || ch == '\t' || ch == '\n' || ch == '\r' || ch == '#') {
      if (in_word) out->push_back(cur);
      cur.clear();
      in_word = false;
      continue;
    }
    cur += ch;
    in_word = true;
  }
  return !quoted;
}

static inline bool expand_config(std::vector<std::string>& args) {
  std::vector<std::string> out;
  if (const char* env = getenv("GPUMON_ARGS")) split_words(env, &out);
  for (size_t i = 0; i < args.size(); i++) {
    if (args[i] != "--config") {
      out.push_back(args[i]);
      continue;
    }
    if (i + 1 >= args.size()) return false;
    FILE* f = fopen(args[++i].c_str(), "r");
    if (!f) {
      fprintf(stderr, "cannot read config file %s\n", args[i].c_str());
      return false;
    }
    std::string text;
    char buf[4096];
    size_t n;
    while ((n = fread(buf, 1, sizeof buf, f)) > 0) text.append(buf, n);
    fclose(f);
    if (!split_words(text, &out)) {
      fprintf(stderr, "unterminated quote in %s\n", args[i].c_str());
      return false;
    }
  }
  args.swap(out);
  return true;
}

// Consumes the connection flags out of args; returns false on a missing value.
static inline bool parse_conn(std::vector<std::string>& args, CliConn* c) {
  if (!expand_config(args)) return false;
  std::vector<std::string> rest;
  for (size_t i = 0; i < args.size(); i++) {
    const std::string& a = args[i];
    auto val = [&](std::string* dst) {
      if (i + 1 >= args.size()) return false;
      *dst = args[++i];
      return true;
    };
    if (a == "-connect" || a == "--connect") { if (!val(&c->connect)) return false; }
    else if (a == "-socket" || a == "--socket") c->is_socket = true;
    else if (a == "--start-hostengine") c->start_hostengine = true;
    else if (a == "--hostengine-path") { if (!val(&c->hostengine)) return false; }
    else if (a == "--backend") { if (!val(&c->backend)) return false; }
    else if (a == "--backend-spec") { if (!val(&c->backend_spec)) return false; }
    else if (a.compare(0, 9, "--health-") == 0) {
      std::string v;
      if (!val(&v)) return false;
      c->health.push_back({a.substr(9), v});
    }
    else rest.push_back(a);
  }
  args.swap(rest);
  return true;
}

static inline gm_status_t cli_init(const CliConn& c) {
  gm_init_params_t p;
  memset(&p, 0, sizeof p);
  p.version = GM_INIT_PARAMS_VERSION;
  p.mode = c.start_hostengine ? GM_MODE_START_HOSTENGINE : (c.connect.empty() ? GM_MODE_EMBEDDED : GM_MODE_STANDALONE);
  p.address = c.connect.empty() ? nullptr : c.connect.c_str();
  p.address_is_unix_socket = c.is_socket ? 1 : 0;
  p.backend = c.backend.empty() ? nullptr : c.backend.c_str();
  p.backend_spec = c.backend_spec.empty() ? nullptr : c.backend_spec.c_str();
  p.hostengine_path = c.hostengine.empty() ? nullptr : c.hostengine.c_str();
  gm_status_t st = gm_init(&p);
  if (st != GM_ST_OK || c.health.empty()) return st;
  gm_health_thresholds_t t;
  memset(&t, 0, sizeof t);
  t.version = GM_HEALTH_THRESHOLDS_VERSION;
  if ((st = gm_health_thresholds_get(&t)) == GM_ST_OK) {
    for (auto& kv : c.health) {
      const double v = atof(kv.second.c_str());
      if (kv.first == "hotspot-warn") t.hotspot_warn_c = (int32_t)v;
      else if (kv.first == "hotspot-fail") t.hotspot_fail_c = (int32_t)v;
      else if (kv.first == "mem-warn") t.mem_warn_c = (int32_t)v;
      else if (kv.first == "mem-fail") t.mem_fail_c = (int32_t)v;
      else if (kv.first == "bad-pages-fail") t.bad_pages_fail = (int32_t)v;
      else if (kv.first == "ce-burst-warn") t.ce_burst_warn = (int32_t)v;
      else if (kv.first == "power-over-cap") t.power_over_cap_warn = v;
      else {
        fprintf(stderr, "unknown health level --health-%s (hotspot-warn, hotspot-fail, mem-warn, mem-fail, bad-pages-fail, ce-burst-warn, power-over-cap)\n",
                kv.first.c_str());
        st = GM_ST_BADPARAM;
      }
    }
    if (st == GM_ST_OK) st = gm_health_thresholds_set(&t);
  }
  if (st != GM_ST_OK) {
    fprintf(stderr, "health levels not accepted: %s\n", gm_strerror(st));
    gm_shutdown();
  }
  return st;
}

#define CLI_CHECK(expr)                                                        \
  do {                                                                         \
    gm_status_t _st = (expr);                                                  \
    if (_st != GM_ST_OK) {                                                     \
      fprintf(stderr, "error: %s: %s\n", #expr, gm_strerror(_st));             \
      gm_shutdown();                                                           \
      return 1;                                                                \
    }                                                                          \
  } while (0)

static inline std::string render_view(int view, uint32_t id, bool json, gm_status_t* st) {
  std::vector<char> buf(1 << 16);
  size_t len = buf.size();
  *st = gm_render_view(view, id, json ? 1 : 0, buf.data(), &len);
  if (*st == GM_ST_INSUFFICIENT_SIZE) {
    buf.resize(len);
    *st = gm_render_view(view, id, json ? 1 : 0, buf.data(), &len);
  }
  return *st == GM_ST_OK ? std::string(buf.data()) : std::string();
}
