// Which processes use a GPU, from /proc alone.
//
// The device library builds its process list from the KFD process table (/sys/class/kfd/kfd/proc), which
// containers and guests often do not have.  The kernel also describes every open DRM file in
// /proc/<pid>/fdinfo/<fd>: for an amdgpu client the card's PCI address (drm-pdev), a client id, its memory by
// placement (drm-memory-vram / -gtt / -cpu, KiB) and the time each engine has spent on its work (drm-engine-gfx,
// -compute, ..., ns).  Those entries are readable for the caller's own processes, and for all processes as root.
// Where the kernel writes no DRM keys (old kernels, some virtual functions), what is left is the open file
// itself: /proc/<pid>/fd/<n> -> /dev/dri/renderD<minor> says that the process holds the card open, not how
// much of it.
//
// Reference: the reference reads the process list from the driver library and names processes from
// /proc/<pid>/comm (bindings/go/nvml/bindings.go:491-582, 637-649).
#pragma once

#include <string>
#include <vector>

#include "backend.h"

namespace gpumon {

enum class ProcSource { NONE, KFD, FDINFO, OPEN_FILES };
const char* proc_source_name(ProcSource s);

struct ProcScan {
  std::vector<ProcSample> procs;
  ProcSource source = ProcSource::NONE;
  int pids_seen = 0;      // numeric directories under the proc root
  int pids_readable = 0;  // of those, the ones whose fd table this process may list
};

// Walks `proc_root` ("/proc") for users of the card at `pci_bdf` whose render node is /dev/dri/renderD<render_minor>
// (either may be unknown: "" / -1; a process is matched by whichever the kernel reports for it).
// Memory of a process that only shows up through its open file is kProcMemUnknown.
void scan_gpu_processes(const std::string& proc_root, const std::string& pci_bdf, int render_minor, ProcScan* out);

constexpr uint64_t kProcMemUnknown = ~0ull;

}  // namespace gpumon
