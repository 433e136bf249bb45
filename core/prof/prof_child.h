// Hardware-counter windows taken by a short-lived child process.
//
// Keeping the compute runtime open inside a monitoring daemon has a standing cost that has nothing to do with the
// counters (one runtime thread in a kernel wait that burns 15 % of a core on some hosts).  This sampler instead
// starts `gpumon-prof-probe --once` every `period_s`: the child opens the runtime, counts one short window per
// counter set, prints one line per GPU and exits.  The daemon pays the child's start-up once per period, which is
// what the period trades against freshness; readers get the last window with its age, and nothing once it is
// older than three periods.
//
// The values are duty-cycle samples (a fraction of a second out of every period), not integrals: the right tool
// for "is the matrix pipe busy on this node", not for accounting.
#pragma once
#include <sys/types.h>

#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <map>
#include <mutex>
#include <string>
#include <thread>

namespace gpumon {

struct ProfWindow {
  double window_s = 0, gpu_active = 0, cu_active = 0, mem_pipe_active = 0, mfma_active = 0, clock_mhz = 0;  // NaN = not measured
  int64_t taken_mono_us = 0;
};

// Parses one "sample pci=... key=value ..." line of the probe; false for any other line.
bool parse_probe_line(const char* line, std::string* pci_bdf, ProfWindow* out);

class ProfChildSampler {
 public:
  // probe: path of gpumon-prof-probe ("" = next to the library this code is in, then $PATH).
  ProfChildSampler(std::string probe, double period_s, int window_ms);
  ~ProfChildSampler();
  void start();
  // Latest window of the GPU with this PCI address (lower case, dddd:bb:dd.f); false when there is none that is
  // younger than three periods.
  bool latest(const std::string& pci_bdf, ProfWindow* out) const;
  uint64_t runs() const { return runs_.load(); }
  uint64_t failures() const { return failures_.load(); }
  double child_cpu_seconds() const;  // user + system time of all children so far
  static std::string default_probe();

 private:
  void loop();
  bool run_once();
  std::string probe_;
  double period_s_;
  int window_ms_;
  mutable std::mutex mu_;
  std::condition_variable cv_;
  std::map<std::string, ProfWindow> latest_;
  std::thread thread_;
  bool stop_ = false, started_ = false, warned_ = false;
  pid_t child_ = -1;
  double child_cpu_s_ = 0;
  std::atomic<uint64_t> runs_{0}, failures_{0};
};

}  // namespace gpumon
