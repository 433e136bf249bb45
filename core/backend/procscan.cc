#include "procscan.h"

#include <dirent.h>
#include <limits.h>
#include <strings.h>
#include <unistd.h>

#include <cstdlib>
#include <cstring>
#include <map>
#include <set>

#include "../common/util.h"

namespace gpumon {

const char* proc_source_name(ProcSource s) {
  switch (s) {
    case ProcSource::KFD: return "kfd";
    case ProcSource::FDINFO: return "fdinfo";
    case ProcSource::OPEN_FILES: return "open-files";
    default: return "none";
  }
}

namespace {

bool all_digits(const char* s) {
  if (!*s) return false;
  for (; *s; s++)
    if (*s < '0' || *s > '9') return false;
  return true;
}

// "drm-memory-vram:\t1048576 KiB" -> bytes; "drm-engine-gfx:\t123 ns" -> ns.  Units other than the ones the
// driver writes today are taken at face value.
uint64_t value_of(const std::string& rest) {
  const char* p = rest.c_str();
  while (*p == ' ' || *p == '\t') p++;
  char* end = nullptr;
  unsigned long long v = strtoull(p, &end, 10);
  while (end && (*end == ' ' || *end == '\t')) end++;
  if (end && strncasecmp(end, "KiB", 3) == 0) v <<= 10;
  else if (end && strncasecmp(end, "MiB", 3) == 0) v <<= 20;
  else if (end && strncasecmp(end, "GiB", 3) == 0) v <<= 30;
  return v;
}

struct Client {  // one DRM client = one open of the device by one process
  uint64_t vram = 0, gtt = 0, cpu = 0, gfx_ns = 0, enc_ns = 0;
};

// Parses one fdinfo file.  Returns true when it describes an amdgpu client; *pdev and *client_id as written.
bool parse_fdinfo(const std::string& text, std::string* pdev, long long* client_id, Client* c) {
  bool amdgpu = false, any_key = false;
  *client_id = -1;
  size_t p = 0;
  while (p < text.size()) {
    size_t nl = text.find('\n', p);
    if (nl == std::string::npos) nl = text.size();
    const std::string line = text.substr(p, nl - p);
    p = nl + 1;
    const size_t colon = line.find(':');
    if (colon == std::string::npos || line.compare(0, 4, "drm-") != 0) continue;
    const std::string key = line.substr(0, colon), rest = line.substr(colon + 1);
    any_key = true;
    if (key == "drm-driver") amdgpu = rest.find("amdgpu") != std::string::npos;
    else if (key == "drm-pdev") {
      size_t b = rest.find_first_not_of(" \t");
      *pdev = b == std::string::npos ? "" : rest.substr(b);
      while (!pdev->empty() && (pdev->back() == ' ' || pdev->back() == '\r')) pdev->pop_back();
    } else if (key == "drm-client-id") *client_id = atoll(rest.c_str());
    else if (key == "drm-memory-vram") c->vram = value_of(rest);
    else if (key == "drm-memory-gtt") c->gtt = value_of(rest);
    else if (key == "drm-memory-cpu") c->cpu = value_of(rest);
    // compute rings carry a CDNA part's kernels; gfx is there on parts that have a graphics ring
    else if (key == "drm-engine-gfx" || key == "drm-engine-compute") c->gfx_ns += value_of(rest);
    else if (key == "drm-engine-enc" || key == "drm-engine-dec") c->enc_ns += value_of(rest);
  }
  return any_key && amdgpu;
}

}  // namespace

void scan_gpu_processes(const std::string& proc_root, const std::string& pci_bdf, int render_minor, ProcScan* out) {
  out->procs.clear();
  out->source = ProcSource::NONE;
  out->pids_seen = out->pids_readable = 0;
  const std::string render = render_minor >= 0 ? "/dev/dri/renderD" + std::to_string(render_minor) : "";
  DIR* root = opendir(proc_root.c_str());
  if (!root) return;
  bool any_fdinfo = false;
  std::vector<ProcSample> holders;  // processes known only by their open file
  while (dirent* pe = readdir(root)) {
    if (!all_digits(pe->d_name)) continue;
    out->pids_seen++;
    const std::string pdir = proc_root + "/" + pe->d_name;
    DIR* fds = opendir((pdir + "/fd").c_str());
    if (!fds) continue;  // not ours to read, or gone already
    out->pids_readable++;
    std::map<long long, Client> clients;  // by DRM client id: a dup'ed or inherited descriptor is the same client
    long long anon = -2;
    bool holds_open = false;
    while (dirent* fe = readdir(fds)) {
      if (!all_digits(fe->d_name)) continue;
      char target[PATH_MAX];
      const ssize_t n = readlink((pdir + "/fd/" + fe->d_name).c_str(), target, sizeof target - 1);
      if (n <= 0) continue;
      target[n] = 0;
      if (strncmp(target, "/dev/dri/", 9) != 0) continue;  // cheap filter: only DRM files are read further
      std::string text, pdev;
      long long id = -1;
      Client c;
      if (read_file(pdir + "/fdinfo/" + fe->d_name, &text) && parse_fdinfo(text, &pdev, &id, &c)) {
        const bool same_card = !pci_bdf.empty() && !pdev.empty() ? strcasecmp(pdev.c_str(), pci_bdf.c_str()) == 0
                                                                 : (!render.empty() && render == target);
        if (!same_card) continue;
        any_fdinfo = true;
        clients[id >= 0 ? id : anon--] = c;
      } else if (!render.empty() && render == target) {
        holds_open = true;
      }
    }
    closedir(fds);
    if (clients.empty() && !holds_open) continue;
    ProcSample p;
    p.pid = (uint32_t)atoll(pe->d_name);
    std::string comm;
    if (read_file(pdir + "/comm", &comm)) {
      while (!comm.empty() && (comm.back() == '\n' || comm.back() == ' ')) comm.pop_back();
      p.name = comm;
    }
    if (!clients.empty()) {
      for (auto& kv : clients) {
        p.vram_bytes += kv.second.vram, p.gtt_bytes += kv.second.gtt, p.cpu_mem_bytes += kv.second.cpu;
        p.gfx_engine_ns += kv.second.gfx_ns, p.enc_engine_ns += kv.second.enc_ns;
      }
      out->procs.push_back(p);
    } else {
      p.vram_bytes = p.gtt_bytes = p.cpu_mem_bytes = kProcMemUnknown;
      holders.push_back(p);
    }
  }
  closedir(root);
  if (any_fdinfo) {
    out->source = ProcSource::FDINFO;
    // a process the kernel describes and one it does not, on one card: keep both, the second without figures
    for (auto& h : holders) out->procs.push_back(h);
  } else if (!holders.empty()) {
    out->source = ProcSource::OPEN_FILES;
    out->procs = holders;
  } else if (out->pids_readable) {
    // nothing found, but the walk itself worked: an empty list from whichever level the kernel offers
    out->source = ProcSource::OPEN_FILES;
  }
}

}  // namespace gpumon
