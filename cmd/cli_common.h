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

// Consumes the connection flags out of args; returns false on a missing value.
static inline bool parse_conn(std::vector<std::string>& args, CliConn* c) {
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
  return gm_init(&p);
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
