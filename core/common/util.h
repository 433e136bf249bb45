// Small shared helpers: clocks, logging, string and file utilities.
#pragma once

#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace gpumon {

int64_t now_us();       // wall clock, microseconds since the epoch
int64_t monotonic_us(); // CLOCK_MONOTONIC microseconds

enum LogLevel { LOG_ERROR = 0, LOG_WARN = 1, LOG_INFO = 2, LOG_DEBUG = 3 };
void set_log_level(int level);
int log_level();
void logf(int level, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

#define GM_LOGE(...) ::gpumon::logf(::gpumon::LOG_ERROR, __VA_ARGS__)
#define GM_LOGW(...) ::gpumon::logf(::gpumon::LOG_WARN, __VA_ARGS__)
#define GM_LOGI(...) ::gpumon::logf(::gpumon::LOG_INFO, __VA_ARGS__)
#define GM_LOGD(...) ::gpumon::logf(::gpumon::LOG_DEBUG, __VA_ARGS__)

// Copies src into a fixed char array, always NUL-terminated.
template <size_t N>
inline void copy_str(char (&dst)[N], const std::string& src) {
  size_t n = src.size() < N - 1 ? src.size() : N - 1;
  memcpy(dst, src.data(), n);
  dst[n] = 0;
}
inline void copy_str(char* dst, size_t cap, const std::string& src) {
  if (!cap) return;
  size_t n = src.size() < cap - 1 ? src.size() : cap - 1;
  memcpy(dst, src.data(), n);
  dst[n] = 0;
}

// Timed condition-variable wait on the monotonic clock.  ThreadSanitizer builds
// route it through the system clock instead: libstdc++ implements steady-clock
// waits with pthread_cond_clockwait, which GCC 11's libtsan does not intercept —
// it then believes the mutex stays locked across the wait and floods the report
// with "double lock" and bogus races.
template <class CV, class Lock, class Rep, class Period, class Pred>
inline bool cv_wait_for(CV& cv, Lock& lock, const std::chrono::duration<Rep, Period>& d, Pred pred) {
#if defined(__SANITIZE_THREAD__)
  return cv.wait_until(lock, std::chrono::system_clock::now() + d, pred);
#else
  return cv.wait_for(lock, d, pred);
#endif
}

std::vector<std::string> split(const std::string& s, char sep);
std::string trim(const std::string& s);
bool read_file(const std::string& path, std::string* out);
// Writes to "<path>.swp"-style temp in the same directory, fsync-free, then rename()s.
bool write_file_atomic(const std::string& path, const char* data, size_t len, int mode = 0644);
std::string strprintf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
// /proc/<pid>/comm without the trailing newline; "" when the process is gone
// (same TOCTOU tolerance as bindings/go/nvml/bindings.go:637-649 of the reference).
std::string process_name(uint32_t pid);

// PCIe per-lane bandwidth in MB/s by generation.  Gen1-4 are the reference's
// table (bindings/go/nvml/nvml.go:314-326); Gen5/6 extend it for MI355X hosts.
int64_t pcie_bandwidth_mbs(unsigned gen, unsigned width);

}  // namespace gpumon
