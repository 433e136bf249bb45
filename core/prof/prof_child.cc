#include "prof_child.h"

#include <dlfcn.h>
#include <fcntl.h>
#include <poll.h>
#include <signal.h>
#include <spawn.h>
#include <sys/resource.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <cerrno>

#include "../common/util.h"

extern char** environ;

namespace gpumon {

namespace {

int64_t mono_us() {
  return std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

bool parse_probe_line(const char* line, std::string* pci_bdf, ProfWindow* out) {
  if (strncmp(line, "sample ", 7) != 0) return false;
  const double nan = std::nan("");
  ProfWindow w;
  w.window_s = 0;
  w.gpu_active = w.cu_active = w.mem_pipe_active = w.mfma_active = w.clock_mhz = nan;
  std::string pci;
  const char* p = line + 7;
  while (*p) {
    while (*p == ' ') p++;
    const char* eq = strchr(p, '=');
    if (!eq) break;
    const std::string key(p, (size_t)(eq - p));
    const char* end = eq + 1;
    while (*end && *end != ' ' && *end != '\n') end++;
    const std::string val(eq + 1, (size_t)(end - eq - 1));
    if (key == "pci") pci = val;
    else {
      char* stop = nullptr;
      const double v = strtod(val.c_str(), &stop);  // "nan" parses to NaN
      const bool ok = stop && stop != val.c_str();
      double* dst = key == "window_s" ? &w.window_s : key == "gpu_active" ? &w.gpu_active : key == "cu_active" ? &w.cu_active
                  : key == "mem_pipe_active" ? &w.mem_pipe_active : key == "mfma_active" ? &w.mfma_active
                  : key == "clock_mhz" ? &w.clock_mhz : nullptr;
      if (dst && ok) *dst = v;
    }
    p = end;
  }
  if (pci.empty()) return false;
  for (auto& c : pci) c = (char)tolower((unsigned char)c);
  for (double* v : {&w.gpu_active, &w.cu_active, &w.mem_pipe_active, &w.mfma_active})
    if (!std::isnan(*v) && (*v < 0 || *v > 100.0)) *v = nan;  // a percentage or nothing
  *pci_bdf = pci;
  *out = w;
  return true;
}

std::string ProfChildSampler::default_probe() {
  if (const char* e = getenv("GPUMON_PROF_PROBE"); e && *e) return e;
  Dl_info info;
  if (dladdr(reinterpret_cast<void*>(&parse_probe_line), &info) && info.dli_fname) {
    std::string self = info.dli_fname;
    const size_t slash = self.rfind('/');
    const std::string dir = slash == std::string::npos ? std::string(".") : self.substr(0, slash);
    for (const char* rel : {"/bin/gpumon-prof-probe", "/gpumon-prof-probe", "/../bin/gpumon-prof-probe"}) {
      const std::string c = dir + rel;
      if (access(c.c_str(), X_OK) == 0) return c;
    }
  }
  return "gpumon-prof-probe";
}

ProfChildSampler::ProfChildSampler(std::string probe, double period_s, int window_ms)
    : probe_(probe.empty() ? default_probe() : std::move(probe)),
      period_s_(period_s < 1.0 ? 1.0 : period_s),
      window_ms_(window_ms < 20 ? 20 : window_ms > 5000 ? 5000 : window_ms) {}

ProfChildSampler::~ProfChildSampler() {
  pid_t child;
  {
    std::lock_guard<std::mutex> l(mu_);
    stop_ = true;
    child = child_;
  }
  cv_.notify_all();
  if (child > 0) kill(child, SIGTERM);  // the exact child; run_once reaps it
  if (thread_.joinable()) thread_.join();
}

void ProfChildSampler::start() {
  std::lock_guard<std::mutex> l(mu_);
  if (started_) return;
  started_ = true;
  thread_ = std::thread([this] { loop(); });
}

bool ProfChildSampler::latest(const std::string& pci_bdf, ProfWindow* out) const {
  std::string key = pci_bdf;
  for (auto& c : key) c = (char)tolower((unsigned char)c);
  std::lock_guard<std::mutex> l(mu_);
  auto it = latest_.find(key);
  if (it == latest_.end()) return false;
  if (mono_us() - it->second.taken_mono_us > (int64_t)(3.0 * period_s_ * 1e6) + 2000000) return false;
  *out = it->second;
  return true;
}

double ProfChildSampler::child_cpu_seconds() const {
  std::lock_guard<std::mutex> l(mu_);
  return child_cpu_s_;
}

void ProfChildSampler::loop() {
  for (;;) {
    const bool ok = run_once();
    runs_++;
    if (!ok) failures_++;
    std::unique_lock<std::mutex> l(mu_);
    // After three failures in a row the child is tried once a minute only: a host without the counters must not
    // pay for a start-up every period.
    const double wait_s = (!ok && failures_.load() >= 3 && failures_.load() == runs_.load()) ? std::max(60.0, period_s_) : period_s_;
    if (cv_wait_for(cv_, l, std::chrono::microseconds((int64_t)(wait_s * 1e6)), [this] { return stop_; })) return;
  }
}

bool ProfChildSampler::run_once() {
  int pfd[2];
  if (pipe2(pfd, O_CLOEXEC) != 0) return false;
  posix_spawn_file_actions_t fa;
  posix_spawn_file_actions_init(&fa);
  posix_spawn_file_actions_adddup2(&fa, pfd[1], 1);
  posix_spawn_file_actions_addopen(&fa, 0, "/dev/null", O_RDONLY, 0);
  const std::string ms = std::to_string(window_ms_);
  std::vector<char*> argv = {const_cast<char*>(probe_.c_str()), const_cast<char*>("--once"), const_cast<char*>("--interval-ms"),
                             const_cast<char*>(ms.c_str()), nullptr};
  pid_t pid = -1;
  const int rc = posix_spawnp(&pid, probe_.c_str(), &fa, nullptr, argv.data(), environ);
  posix_spawn_file_actions_destroy(&fa);
  close(pfd[1]);
  if (rc != 0) {
    close(pfd[0]);
    if (!warned_) GM_LOGW("profiling counters: cannot start %s: %s", probe_.c_str(), strerror(rc));
    warned_ = true;
    return false;
  }
  {
    std::lock_guard<std::mutex> l(mu_);
    child_ = pid;
  }
  // Read until the child closes its end, it overruns its time, or the sampler stops.
  std::string text;
  const int64_t deadline = mono_us() + (int64_t)window_ms_ * 1000 * 8 + 20000000;  // start-up of the runtime dominates
  bool timed_out = false;
  for (;;) {
    struct pollfd p = {pfd[0], POLLIN, 0};
    const int pr = poll(&p, 1, 200);
    if (pr > 0) {
      char buf[2048];
      const ssize_t n = read(pfd[0], buf, sizeof buf);
      if (n <= 0) break;
      text.append(buf, (size_t)n);
      if (text.size() > (1u << 20)) break;
    }
    bool stopping;
    {
      std::lock_guard<std::mutex> l(mu_);
      stopping = stop_;
    }
    if (stopping || mono_us() > deadline) {
      timed_out = !stopping;
      kill(pid, SIGKILL);
      break;
    }
  }
  close(pfd[0]);
  int st = 0;
  struct rusage ru;
  memset(&ru, 0, sizeof ru);
  while (wait4(pid, &st, 0, &ru) < 0 && errno == EINTR) {
  }
  const int64_t now = mono_us();
  bool any = false;
  {
    std::lock_guard<std::mutex> l(mu_);
    child_ = -1;
    child_cpu_s_ += ru.ru_utime.tv_sec + ru.ru_stime.tv_sec + (ru.ru_utime.tv_usec + ru.ru_stime.tv_usec) * 1e-6;
    size_t pos = 0;
    while (pos < text.size()) {
      size_t nl = text.find('\n', pos);
      if (nl == std::string::npos) nl = text.size();
      const std::string line = text.substr(pos, nl - pos);
      pos = nl + 1;
      std::string pci;
      ProfWindow w;
      if (!parse_probe_line(line.c_str(), &pci, &w)) continue;
      w.taken_mono_us = now;
      latest_[pci] = w;
      any = true;
    }
  }
  const bool ok = any && !timed_out && WIFEXITED(st) && WEXITSTATUS(st) == 0;
  if (!ok && !warned_) {
    GM_LOGW("profiling counters: %s --once %s", probe_.c_str(),
            timed_out ? "overran its time and was killed" : any ? "ended with an error" : "gave no sample");
    warned_ = true;
  }
  return ok;
}

}  // namespace gpumon
