#include "util.h"

#include <atomic>
#include <cerrno>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

namespace gpumon {

int64_t now_us() {
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

int64_t monotonic_us() {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

static std::atomic<int> g_level{[] {
  const char* e = getenv("GPUMON_LOG");
  return e ? atoi(e) : (int)LOG_WARN;
}()};

void set_log_level(int level) { g_level.store(level); }
int log_level() { return g_level.load(); }

void logf(int level, const char* fmt, ...) {
  if (level > g_level.load(std::memory_order_relaxed)) return;
  static const char* names[] = {"E", "W", "I", "D"};
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  struct timeval tv;
  gettimeofday(&tv, nullptr);
  struct tm tm;
  localtime_r(&tv.tv_sec, &tm);
  fprintf(stderr, "%s%02d%02d %02d:%02d:%02d.%06ld gpumon] %s\n", names[level & 3], tm.tm_mon + 1,
          tm.tm_mday, tm.tm_hour, tm.tm_min, tm.tm_sec, (long)tv.tv_usec, buf);
}

std::vector<std::string> split(const std::string& s, char sep) {
  std::vector<std::string> out;
  size_t start = 0;
  while (true) {
    size_t p = s.find(sep, start);
    if (p == std::string::npos) {
      out.push_back(s.substr(start));
      break;
    }
    out.push_back(s.substr(start, p - start));
    start = p + 1;
  }
  return out;
}

std::string trim(const std::string& s) {
  size_t b = 0, e = s.size();
  while (b < e && isspace((unsigned char)s[b])) b++;
  while (e > b && isspace((unsigned char)s[e - 1])) e--;
  return s.substr(b, e - b);
}

bool read_file(const std::string& path, std::string* out) {
  int fd = open(path.c_str(), O_RDONLY | O_CLOEXEC);
  if (fd < 0) return false;
  out->clear();
  char buf[4096];
  ssize_t n;
  while ((n = read(fd, buf, sizeof buf)) > 0) out->append(buf, (size_t)n);
  close(fd);
  return n == 0;
}

bool write_file_atomic(const std::string& path, const char* data, size_t len, int mode) {
  std::string tmp = path + ".swp";
  int fd = open(tmp.c_str(), O_WRONLY | O_CREAT | O_TRUNC | O_CLOEXEC, mode);
  if (fd < 0) return false;
  size_t off = 0;
  while (off < len) {
    ssize_t n = write(fd, data + off, len - off);
    if (n < 0) {
      if (errno == EINTR) continue;
      close(fd);
      unlink(tmp.c_str());
      return false;
    }
    off += (size_t)n;
  }
  fchmod(fd, mode);
  close(fd);
  if (rename(tmp.c_str(), path.c_str()) != 0) {
    unlink(tmp.c_str());
    return false;
  }
  return true;
}

std::string strprintf(const char* fmt, ...) {
  char buf[2048];
  va_list ap;
  va_start(ap, fmt);
  int n = vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (n < 0) return std::string();
  if ((size_t)n < sizeof buf) return std::string(buf, (size_t)n);
  std::string big((size_t)n + 1, 0);
  va_start(ap, fmt);
  vsnprintf(&big[0], big.size(), fmt, ap);
  va_end(ap);
  big.resize((size_t)n);
  return big;
}

std::string process_name(uint32_t pid) {
  std::string s;
  if (!read_file(strprintf("/proc/%u/comm", pid), &s)) return std::string();
  return trim(s);
}

int64_t pcie_bandwidth_mbs(unsigned gen, unsigned width) {
  static const int64_t per_lane[] = {0, 250, 500, 985, 1969, 3938, 7563};
  if (gen == 0 || gen >= sizeof per_lane / sizeof per_lane[0] || width == 0) return -1;
  return per_lane[gen] * (int64_t)width;
}

}  // namespace gpumon
