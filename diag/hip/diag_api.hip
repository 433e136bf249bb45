// Device bookkeeping and error reporting of the diagnostics library.
#include "diag_common.h"

namespace gmd {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

}  // namespace gmd

extern "C" {

const char* gmd_last_error(void) { return gmd::g_err; }

int gmd_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int gmd_device_info(int device, gmd_device_info_t* out) {
  if (!out) return GMD_ERR_BADPARAM;
  memset(out, 0, sizeof *out);
  hipDeviceProp_t p;
  GMD_HIP(hipGetDeviceProperties(&p, device));
  GMD_HIP(hipSetDevice(device));
  snprintf(out->name, sizeof out->name, "%s", p.name);
  snprintf(out->gcn_arch, sizeof out->gcn_arch, "%s", p.gcnArchName);
  size_t free_b = 0, total_b = 0;
  GMD_HIP(hipMemGetInfo(&free_b, &total_b));
  out->total_mem = total_b;
  out->free_mem = free_b;
  out->compute_units = p.multiProcessorCount;
  out->clock_mhz = p.clockRate / 1000;
  out->mem_clock_mhz = p.memoryClockRate / 1000;
  out->l2_bytes = p.l2CacheSize;
  out->lds_per_cu = (int)p.maxSharedMemoryPerMultiProcessor;
  snprintf(out->pci_bdf, sizeof out->pci_bdf, "%04x:%02x:%02x.0", p.pciDomainID, p.pciBusID, p.pciDeviceID);
  return GMD_OK;
}

}  // extern "C"
