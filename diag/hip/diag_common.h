// Helpers shared by the diagnostic translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "gpumon_diag.h"

namespace gmd {

void set_error(const char* fmt, ...) __attribute__((format(printf, 1, 2)));

#define GMD_HIP(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) {                                                                    \
      ::gmd::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return _e == hipErrorOutOfMemory ? GMD_ERR_NOMEM : GMD_ERR_HIP;                          \
    }                                                                                          \
  } while (0)

constexpr int kCUs = 256;  // MI355X: 8 XCDs x 32 CUs

// Grid for a memory-bound grid-stride kernel: enough 256-thread blocks to give
// every CU eight resident blocks, never more than the work needs.
inline unsigned stream_grid(uint64_t work_items, unsigned block, unsigned items_per_thread) {
  uint64_t need = (work_items + (uint64_t)block * items_per_thread - 1) / ((uint64_t)block * items_per_thread);
  uint64_t cap = (uint64_t)kCUs * 8;
  return (unsigned)std::max<uint64_t>(1, std::min(need, cap));
}

// Launch plan for a streaming kernel in which every workgroup walks ONE contiguous
// span of the buffer: `pass` elements are handled by one unrolled workgroup
// iteration; the span is a whole number of passes; at most `max_grid` workgroups.
// Measured on MI355X, many short spans (128 workgroups per CU) beat a grid-stride
// walk by 10-15 % on kernels that store.
inline void span_plan(uint64_t n, uint64_t pass, unsigned max_grid, unsigned* grid, uint64_t* span) {
  uint64_t passes = (n + pass - 1) / pass;
  uint64_t g = std::max<uint64_t>(1, std::min<uint64_t>(passes, max_grid));
  uint64_t per = (passes + g - 1) / g;
  *span = per * pass;
  *grid = (unsigned)std::max<uint64_t>(1, (n + *span - 1) / *span);
}

struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
  ~DeviceBuffer() {
    if (p) (void)hipFree(p);
  }
  int alloc(size_t n) {
    GMD_HIP(hipMalloc(&p, n));
    bytes = n;
    return GMD_OK;
  }
};

}  // namespace gmd
