// Device-wide hardware counters for the profiling field (of the "-p" metric set) that the
// device library has no source for: MFMA pipe activity, the analogue of tensor-pipe activity.
//
// This is a separate shared library (build/libgpumon_prof.so) because it is the only part
// of the monitoring stack that opens the compute runtime (HSA) and programs performance
// counters through rocprofiler-sdk's device counting service; libgpumon.so loads it with
// dlopen only when the operator asks for it (GPUMON_PROF_COUNTERS=1 / --prof-counters),
// and every field it feeds stays blank when it is absent or refused.  It belongs in a
// monitoring process: the registration has to precede the start of HSA, and the HIP runtime
// of a process that has opened the plugin no longer finds a device (measured, ROCm 7.2).
//
// Reference: the DCGM "DCP" fields 1001-1005 the exporter passes through with -p
// (exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:80-83,179-187).
#pragma once
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMP_API __attribute__((visibility("default")))

typedef struct gmp_agent_info_t {
  uint32_t pci_domain;
  uint32_t pci_location;  // bus << 8 | device << 3 | function
  uint32_t cu_count;
  uint32_t simd_count;
  char name[64];
} gmp_agent_info_t;

// Percent of wall-clock time over the window since the previous gmp_sample of the same
// agent; NAN for a value the counters of this device cannot give.
typedef struct gmp_sample_t {
  double window_s;     // length of the window
  double gpu_active;   // graphics engine busy (GRBM_GUI_ACTIVE / GRBM_COUNT)
  double mfma_active;  // matrix pipe busy, averaged over SIMDs and over all cycles of the window
  double clock_mhz;    // shader clock the device held (GRBM_COUNT / window)
  double raw[8];       // folded counter values: GRBM_COUNT, GRBM_GUI_ACTIVE, SQ_VALU_MFMA_BUSY_CYCLES, then the $GPUMON_PROF_EXTRA ones
} gmp_sample_t;

// Registers with rocprofiler-sdk, opens HSA, sets up one counting context per GPU agent.
// Returns the number of GPU agents (>= 0) or a negative value with a message in err.
GMP_API int gmp_open(char* err, int errlen);
GMP_API int gmp_agent_info(int agent, gmp_agent_info_t* out);
// First call for an agent starts its counters and returns 1 (no window yet); later calls
// return 0 and fill *out, or a negative value with a message from gmp_last_error().
GMP_API int gmp_sample(int agent, gmp_sample_t* out);
GMP_API const char* gmp_last_error(void);
GMP_API void gmp_close(void);

#ifdef __cplusplus
}
#endif
