// gpumon-hostengine — the standalone host engine daemon (the nv-hostengine
// role: bindings/go/dcgm/admin.go:149-209, dcgm-exporter:39-42,198-217).
//
//   --domain-socket PATH   serve clients on a unix socket
//   -b ADDR / --port N     serve clients on TCP (default 127.0.0.1:5555)
//   --http ADDR            also serve REST + /metrics (e.g. :8070)
//   --prof-counters[=child] matrix-pipe, CU and memory-pipeline activity from hardware counters: in this process
//                          (libgpumon_prof.so; = GPUMON_PROF_COUNTERS=1) or sampled by a short-lived probe child
//                          every --prof-period seconds (= GPUMON_PROF_COUNTERS=child, GPUMON_PROF_PERIOD_S)
//   --backend NAME --backend-spec SPEC
//   --pid-file FILE        default /run/gpumon-hostengine.pid when writable, else /tmp
//   --daemon               detach from the terminal
//   --term                 stop the daemon named by the pid file
#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include "cli_common.h"

static std::string default_pid_file() {
  return access("/run", W_OK) == 0 ? "/run/gpumon-hostengine.pid" : "/tmp/gpumon-hostengine.pid";
}

int main(int argc, char** argv) {
  std::vector<std::string> args(argv + 1, argv + argc);
  CliConn conn;
  if (!parse_conn(args, &conn)) return 2;
  std::string sock, bind_addr = "127.0.0.1", http, pid_file = default_pid_file();
  int port = 5555;
  bool term = false, daemonize = false, pid_file_set = false;
  for (size_t i = 0; i < args.size(); i++) {
    const std::string& a = args[i];
    auto val = [&](std::string* dst) {
      if (i + 1 >= args.size()) {
        fprintf(stderr, "missing value for %s\n", a.c_str());
        exit(2);
      }
      *dst = args[++i];
    };
    std::string tmp;
    if (a == "--domain-socket" || a == "-d") val(&sock);
    else if (a == "-b" || a == "--bind") val(&bind_addr);
    else if (a == "--port" || a == "-p") { val(&tmp); port = atoi(tmp.c_str()); }
    else if (a == "--http") val(&http);
    else if (a == "--prof-counters") setenv("GPUMON_PROF_COUNTERS", "1", 1);
    else if (a == "--prof-counters=child") setenv("GPUMON_PROF_COUNTERS", "child", 1);
    else if (a == "--prof-period") { val(&tmp); setenv("GPUMON_PROF_PERIOD_S", tmp.c_str(), 1); }
    else if (a == "--pid-file") { val(&pid_file); pid_file_set = true; }
    else if (a == "--daemon") daemonize = true;
    else if (a == "-n") daemonize = false;
    else if (a == "--term" || a == "-t") term = true;
    else {
      fprintf(stderr, "usage: gpumon-hostengine [--domain-socket PATH | -b ADDR --port N] [--http ADDR] [--backend B] [--daemon] [--term]\n");
      return 2;
    }
  }
  if (term) {
    FILE* f = fopen(pid_file.c_str(), "r");
    long pid = 0;
    if (!f || fscanf(f, "%ld", &pid) != 1 || pid <= 1) {
      fprintf(stderr, "no running host engine found (%s)\n", pid_file.c_str());
      if (f) fclose(f);
      return 1;
    }
    fclose(f);
    // A pid file outlives its process, and pids are reused: only a process that is a host engine is signalled.
    {
      char path[64], comm[64] = "";
      snprintf(path, sizeof path, "/proc/%ld/comm", pid);
      FILE* c = fopen(path, "r");
      if (!c || !fgets(comm, sizeof comm, c)) {
        fprintf(stderr, "no running host engine found (pid %ld of %s is gone)\n", pid, pid_file.c_str());
        if (c) fclose(c);
        return 1;
      }
      fclose(c);
      comm[strcspn(comm, "\n")] = 0;
      // the kernel keeps 15 characters of the name: "gpumon-hostengi"
      if (strncmp(comm, "gpumon-hostengine", 15) != 0) {
        fprintf(stderr, "pid %ld (%s) is not a host engine: %s is stale, nothing signalled\n", pid, comm, pid_file.c_str());
        return 1;
      }
    }
    if (kill((pid_t)pid, SIGTERM) != 0) {
      perror("kill");
      return 1;
    }
    for (int i = 0; i < 100; i++) {  // 10 s grace, then force
      // Gone, or a zombie its parent has not reaped yet: either way it has stopped.
      bool stopped = kill((pid_t)pid, 0) != 0;
      if (!stopped) {
        char path[64], state = 0;
        snprintf(path, sizeof path, "/proc/%ld/stat", pid);
        if (FILE* s = fopen(path, "r")) {
          if (fscanf(s, "%*d (%*[^)]) %c", &state) == 1 && state == 'Z') stopped = true;
          fclose(s);
        }
      }
      if (stopped) {
        unlink(pid_file.c_str());
        printf("Host engine successfully terminated.\n");
        return 0;
      }
      usleep(100000);
    }
    fprintf(stderr, "Could not stop the host engine; killing it\n");
    kill((pid_t)pid, SIGKILL);
    unlink(pid_file.c_str());
    return 1;
  }
  if (daemonize) {
    pid_t p = fork();
    if (p < 0) return 1;
    if (p > 0) return 0;
    setsid();
    int nul = open("/dev/null", O_RDWR);
    if (nul >= 0) {
      dup2(nul, 0);
      dup2(nul, 1);
      close(nul);
    }
  }
  install_signals();
  conn.connect.clear();  // a host engine is always embedded
  conn.start_hostengine = false;
  gm_status_t st = cli_init(conn);
  if (st != GM_ST_OK) {
    fprintf(stderr, "error: cannot initialise the engine: %s\n", gm_strerror(st));
    return 1;
  }
  std::string address = sock.empty() ? bind_addr + ":" + std::to_string(port) : sock;
  st = gm_server_start(address.c_str(), sock.empty() ? 0 : 1);
  if (st != GM_ST_OK) {
    fprintf(stderr, "error: cannot serve on %s: %s\n", address.c_str(), gm_strerror(st));
    gm_shutdown();
    return 1;
  }
  // A spawned child (domain socket, no explicit pid file) leaves the shared pid file alone.
  bool write_pid = sock.empty() || pid_file_set;
  if (write_pid) {
    FILE* f = fopen(pid_file.c_str(), "w");
    if (f) {
      fprintf(f, "%d\n", (int)getpid());
      fclose(f);
    }
  }
  gm_exporter_t ex = 0;
  bool have_http = false;
  if (!http.empty()) {
    gm_exporter_config_t cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.version = GM_EXPORTER_CONFIG_VERSION;
    cfg.interval_ms = 1000;
    cfg.listen = http.c_str();
    cfg.rest_api = 1;
    cfg.self_metrics = 1;
    cfg.amd_extras = 1;
    cfg.health = 1;  // the daemon's /metrics carries the health verdicts and policy violation counts too
    if (gm_exporter_create(&cfg, &ex) == GM_ST_OK && gm_exporter_start(ex) == GM_ST_OK) have_http = true;
    else fprintf(stderr, "warning: cannot serve HTTP on %s\n", http.c_str());
  }
  fprintf(stderr, "Host engine listening on %s%s\n", sock.empty() ? "" : "unix:", address.c_str());
  while (!g_stop) {
    usleep(100000);
    // A spawned child follows its parent: exit when re-parented to init.
    if (!sock.empty() && !daemonize && getppid() == 1) break;
  }
  if (have_http) gm_exporter_destroy(ex);
  gm_server_stop();
  gm_shutdown();
  if (write_pid) unlink(pid_file.c_str());
  return 0;
}
