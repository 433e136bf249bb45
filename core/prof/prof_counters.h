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

// The GRBM block has two counter slots.  One holds GRBM_COUNT (the clock every ratio is taken over); the other
// rotates from window to window through a list of device-wide busy signals ($GPUMON_PROF_GRBM, default
// GRBM_GUI_ACTIVE,GRBM_SPI_BUSY,GRBM_TA_BUSY), so a sample carries ONE of gpu_active / cu_active /
// mem_pipe_active and NAN for the others; the matrix-pipe counter lives in the SQ block and is in every window.
//
// Percent of wall-clock time over the window since the previous gmp_sample of the same agent; NAN for a value
// this window did not measure or the counters of this device cannot give.
#define GMP_ABI_VERSION 2
typedef struct gmp_sample_t {
  double window_s;         // length of the window
  double gpu_active;       // graphics engine busy (GRBM_GUI_ACTIVE / GRBM_COUNT)
  double mfma_active;      // matrix pipe busy, averaged over SIMDs and over all cycles of the window
  double clock_mhz;        // shader clock the device held (GRBM_COUNT / window)
  double cu_active;        // wave launcher / shader arrays busy (GRBM_SPI_BUSY / GRBM_COUNT, mean over the counter's instances)
  double mem_pipe_active;  // vector-memory address units busy (GRBM_TA_BUSY / GRBM_COUNT, mean over instances)
  double grbm_busy;        // this window's rotating counter as a percentage, whatever it was
  char grbm_name[32];      // ... and its name
  int32_t grbm_instances;  // instances the rotating counter reported (one per XCC on MI355X)
  int32_t reserved;
  double raw[8];           // folded counter values: GRBM_COUNT, the rotating GRBM counter (summed over instances), SQ_VALU_MFMA_BUSY_CYCLES, then the $GPUMON_PROF_EXTRA ones
} gmp_sample_t;

// Registers with rocprofiler-sdk, opens HSA, sets up one counting context per GPU agent.
// Returns the number of GPU agents (>= 0) or a negative value with a message in err.
GMP_API int gmp_abi_version(void);  // GMP_ABI_VERSION the library was built with
GMP_API int gmp_open(char* err, int errlen);
// Number of counter sets the windows rotate through (valid after gmp_open).
GMP_API int gmp_set_count(void);
GMP_API int gmp_agent_info(int agent, gmp_agent_info_t* out);
// First call for an agent starts its counters and returns 1 (no window yet); later calls
// return 0 and fill *out, or a negative value with a message from gmp_last_error().
GMP_API int gmp_sample(int agent, gmp_sample_t* out);
GMP_API const char* gmp_last_error(void);
GMP_API void gmp_close(void);

#ifdef __cplusplus
}
#endif
