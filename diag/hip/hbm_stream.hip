// HBM stream-bandwidth diagnostic for MI355X (gfx950).
//
// Six STREAM-style kernels.  Every lane moves 16 bytes per access (one
// global_load/store_dwordx4, 1 KiB per wave-instruction — the widest coalesced
// access the memory pipeline has).  No kernel loops: the kernels that store give
// every lane one access (onepass_kernel), the read kernel kUnroll independent loads
// (read_kernel), and the in-order dispatch of a very large grid does the
// traversal.  Buffers are sized far
// beyond the 256 MiB Infinity Cache by the caller so the number reported is
// HBM3E, not L3.  Non-temporal accesses keep the once-touched stream from
// displacing lines in L2 / Infinity Cache; measured on MI355X they are worth
// +12 % on reads (7.2 TB/s), +4-8 % on copies and triads (6.7 / 6.6 TB/s) and
// cost 2-4 % on pure stores (7.1 TB/s plain).
#include "diag_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kBlock = 256;
constexpr int kUnroll = 4;

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// The kernels that store run as ONE access per lane and nothing else: workgroup w
// touches elements [w*BS, (w+1)*BS) and exits.  The dispatcher hands workgroups out
// in order, so the addresses in flight form one compact window that slides through
// the buffers, and reads and writes of the same few DRAM pages arrive together.
// Measured on MI355X (4 GiB buffers, TB/s; scripts/stream_sweep.hip):
//   copy   5.5 (128 KiB spans) -> 6.5 (256 lanes) -> 6.7 (64 lanes), non-temporal
//   triad  5.6                 -> 6.4             -> 6.6
//   write  6.05                -> 6.85 (256 lanes) -> 7.1 with 4 workgroups per CU
// A pure store stream is dispatch-bound with 64-lane workgroups (4.8) and likes fewer
// waves in flight: 40 KiB of (unused) dynamic LDS caps it at 4 workgroups per CU.
template <int OP, bool NT, int BS>
__global__ __launch_bounds__(BS) void onepass_kernel(f32x4* __restrict__ a, f32x4* __restrict__ b, f32x4* __restrict__ c,
                                                     float s, uint64_t n4) {
  const uint64_t i = (uint64_t)blockIdx.x * BS + threadIdx.x;
  if (i >= n4) return;
  if constexpr (OP == GMD_STREAM_COPY) st<NT>(c + i, ld<NT>(a + i));
  if constexpr (OP == GMD_STREAM_SCALE) st<NT>(b + i, s * ld<NT>(c + i));
  if constexpr (OP == GMD_STREAM_ADD) st<NT>(c + i, ld<NT>(a + i) + ld<NT>(b + i));
  if constexpr (OP == GMD_STREAM_TRIAD) st<NT>(a + i, ld<NT>(b + i) + s * ld<NT>(c + i));
  if constexpr (OP == GMD_STREAM_WRITE) st<NT>(a + i, f32x4{s, s, s, s});
}
constexpr int kMoveBlock = 64;            // copy / scale / add / triad
constexpr int kStoreBlock = 256;          // write
constexpr int kStoreLdsBytes = 40 * 1024; // 160 KiB / 40 KiB = 4 workgroups (16 waves) per CU

template <int OP, bool NT>
unsigned launch_onepass(f32x4* a, f32x4* b, f32x4* c, float s, uint64_t n4, hipStream_t st) {
  constexpr int BS = OP == GMD_STREAM_WRITE ? kStoreBlock : kMoveBlock;
  // A launch holds fewer than 2^32 work-items: 32 GiB of 16-byte elements per launch.
  constexpr uint64_t kPerLaunch = 1ull << 31;
  unsigned grid = 0;
  for (uint64_t off = 0; off < n4; off += kPerLaunch) {
    const uint64_t n = std::min(kPerLaunch, n4 - off);
    grid = (unsigned)((n + BS - 1) / BS);
    onepass_kernel<OP, NT, BS><<<grid, BS, OP == GMD_STREAM_WRITE ? kStoreLdsBytes : 0, st>>>(
        a ? a + off : a, b ? b + off : b, c ? c + off : c, s, n);
  }
  return grid;
}

// The read stream in the same form: one unrolled pass (kUnroll independent 16-byte loads per lane, 16 KiB per
// workgroup) and out.  The loads must stay live, so every workgroup adds its sum into one of
// GMD_STREAM_SINK_FLOATS slots (one float atomic per 16 KiB read: far below the atomic rate).
// Measured: 7.0 TB/s with 128 KiB spans -> 7.25 TB/s.
static_assert((GMD_STREAM_SINK_FLOATS & (GMD_STREAM_SINK_FLOATS - 1)) == 0, "slot = workgroup & (slots - 1)");
template <bool NT>
__global__ __launch_bounds__(kBlock) void read_kernel(const f32x4* __restrict__ a, uint64_t n4, float* __restrict__ sink) {
  const uint64_t base = (uint64_t)blockIdx.x * (kBlock * kUnroll) + threadIdx.x;
  f32x4 x[kUnroll];
#pragma unroll
  for (int u = 0; u < kUnroll; u++) {
    const uint64_t j = base + (uint64_t)u * kBlock;
    x[u] = j < n4 ? ld<NT>(a + j) : f32x4{0.f, 0.f, 0.f, 0.f};
  }
  f32x4 acc = (x[0] + x[1]) + (x[2] + x[3]);
  static_assert(kUnroll == 4, "the sum above is written for four loads");
  float t = acc.x + acc.y + acc.z + acc.w;
  for (int off = 32; off > 0; off >>= 1) t += __shfl_down(t, off, 64);
  __shared__ float part[kBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float r = 0;
    for (int w = 0; w < kBlock / 64; w++) r += part[w];
    atomicAdd(sink + (blockIdx.x & (GMD_STREAM_SINK_FLOATS - 1)), r);
  }
}

template <bool NT>
unsigned launch_read(const f32x4* a, uint64_t n4, float* sink, hipStream_t st) {
  constexpr uint64_t kPass = (uint64_t)kBlock * kUnroll;
  constexpr uint64_t kPerLaunch = kPass << 23;  // grid x 256 work-items stays under 2^32: 128 GiB per launch
  unsigned grid = 0;
  for (uint64_t off = 0; off < n4; off += kPerLaunch) {
    const uint64_t n = std::min(kPerLaunch, n4 - off);
    grid = (unsigned)((n + kPass - 1) / kPass);
    read_kernel<NT><<<grid, kBlock, 0, st>>>(a + off, n, sink);
  }
  return grid;
}

__global__ void init_kernel(f32x4* a, f32x4* b, f32x4* c, uint64_t n4) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
    a[i] = f32x4{1.f, 1.f, 1.f, 1.f};
    b[i] = f32x4{2.f, 2.f, 2.f, 2.f};
    c[i] = f32x4{0.5f, 0.5f, 0.5f, 0.5f};
  }
}

template <bool NT>
unsigned launch(int kernel, f32x4* a, f32x4* b, f32x4* c, float s, uint64_t n4, float* sink, hipStream_t st) {
  switch (kernel) {
    case GMD_STREAM_COPY: return launch_onepass<GMD_STREAM_COPY, NT>(a, b, c, s, n4, st);
    case GMD_STREAM_SCALE: return launch_onepass<GMD_STREAM_SCALE, NT>(a, b, c, s, n4, st);
    case GMD_STREAM_ADD: return launch_onepass<GMD_STREAM_ADD, NT>(a, b, c, s, n4, st);
    case GMD_STREAM_TRIAD: return launch_onepass<GMD_STREAM_TRIAD, NT>(a, b, c, s, n4, st);
    case GMD_STREAM_WRITE: return launch_onepass<GMD_STREAM_WRITE, NT>(a, b, c, s, n4, st);
    case GMD_STREAM_READ: return launch_read<NT>(a, n4, sink, st);
  }
  return 0;
}

// nontemporal: 0 = plain, 1 = non-temporal, 2 = the faster policy measured per kernel
// (streams that read benefit from nt loads; a pure store stream does not).
bool use_nt(int kernel, int nontemporal) {
  if (nontemporal == 2) return kernel != GMD_STREAM_WRITE;
  return nontemporal != 0;
}

int bytes_factor(int kernel) {
  switch (kernel) {
    case GMD_STREAM_COPY:
    case GMD_STREAM_SCALE: return 2;
    case GMD_STREAM_ADD:
    case GMD_STREAM_TRIAD: return 3;
    default: return 1;
  }
}

}  // namespace

extern "C" {

int gmd_hbm_stream_launch(int kernel, void* a, void* b, void* c, float scalar, uint64_t n_floats, int nontemporal,
                          void* hip_stream) {
  if (kernel < 0 || kernel >= GMD_STREAM_KERNELS || (n_floats & 3) || !a) return GMD_ERR_BADPARAM;
  if ((kernel == GMD_STREAM_ADD || kernel == GMD_STREAM_TRIAD) && (!b || !c)) return GMD_ERR_BADPARAM;
  if ((kernel == GMD_STREAM_COPY) && !c) return GMD_ERR_BADPARAM;
  if ((kernel == GMD_STREAM_SCALE) && (!b || !c)) return GMD_ERR_BADPARAM;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) return GMD_ERR_BADPARAM;
  hipStream_t st = (hipStream_t)hip_stream;
  // READ adds every workgroup's sum into one of the GMD_STREAM_SINK_FLOATS slots of a sink: the caller's
  // `c` when given (fully asynchronous; zero it first if the sums matter), else a temporary (synchronises).
  float* sink = nullptr;
  bool own_sink = false;
  if (kernel == GMD_STREAM_READ) {
    sink = (float*)c;
    if (!sink) {
      GMD_HIP(hipMalloc((void**)&sink, sizeof(float) * GMD_STREAM_SINK_FLOATS));
      own_sink = true;
    }
  }
  if (use_nt(kernel, nontemporal)) launch<true>(kernel, (f32x4*)a, (f32x4*)b, (f32x4*)c, scalar, n_floats / 4, sink, st);
  else launch<false>(kernel, (f32x4*)a, (f32x4*)b, (f32x4*)c, scalar, n_floats / 4, sink, st);
  hipError_t e = hipGetLastError();
  if (own_sink) {
    (void)hipStreamSynchronize(st);
    (void)hipFree(sink);
  }
  if (e != hipSuccess) {
    gmd::set_error("stream kernel launch failed: %s", hipGetErrorString(e));
    return GMD_ERR_HIP;
  }
  return GMD_OK;
}

int gmd_hbm_stream(int device, int kernel, uint64_t bytes_per_buffer, int iters, int nontemporal,
                   gmd_stream_result_t* out) {
  if (!out || kernel < 0 || kernel >= GMD_STREAM_KERNELS || iters < 1 || bytes_per_buffer < 4096) return GMD_ERR_BADPARAM;
  memset(out, 0, sizeof *out);
  GMD_HIP(hipSetDevice(device));
  const uint64_t n4 = bytes_per_buffer / 16;
  const uint64_t bytes = n4 * 16;
  gmd::DeviceBuffer a, b, c, sink;
  int rc;
  if ((rc = a.alloc(bytes)) || (rc = b.alloc(bytes)) || (rc = c.alloc(bytes)) ||
      (rc = sink.alloc(sizeof(float) * GMD_STREAM_SINK_FLOATS)))
    return rc;
  init_kernel<<<gmd::kCUs * 8, 256>>>((f32x4*)a.p, (f32x4*)b.p, (f32x4*)c.p, n4);
  GMD_HIP(hipGetLastError());
  GMD_HIP(hipDeviceSynchronize());
  const float s = 3.0f;
  unsigned grid = 0;
  auto run = [&]() {
    if (use_nt(kernel, nontemporal)) grid = launch<true>(kernel, (f32x4*)a.p, (f32x4*)b.p, (f32x4*)c.p, s, n4, (float*)sink.p, 0);
    else grid = launch<false>(kernel, (f32x4*)a.p, (f32x4*)b.p, (f32x4*)c.p, s, n4, (float*)sink.p, 0);
  };
  GMD_HIP(hipMemset(sink.p, 0, sizeof(float) * GMD_STREAM_SINK_FLOATS));
  run();  // warm-up: page tables, clocks
  GMD_HIP(hipGetLastError());
  GMD_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  GMD_HIP(hipEventCreate(&e0));
  GMD_HIP(hipEventCreate(&e1));
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; i++) {
    // Triad and scale rewrite their own inputs; re-seed so values stay finite and checkable.
    if (kernel == GMD_STREAM_TRIAD || kernel == GMD_STREAM_SCALE) {
      init_kernel<<<gmd::kCUs * 8, 256>>>((f32x4*)a.p, (f32x4*)b.p, (f32x4*)c.p, n4);
    }
    GMD_HIP(hipEventRecord(e0, 0));
    run();
    GMD_HIP(hipEventRecord(e1, 0));
    GMD_HIP(hipEventSynchronize(e1));
    GMD_HIP(hipEventElapsedTime(&ms[i], e0, e1));
  }
  GMD_HIP(hipGetLastError());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);

  // Verify a spread of elements against the closed-form expectation.
  float expect_val = 0;
  const void* check = nullptr;
  switch (kernel) {
    case GMD_STREAM_COPY: expect_val = 1.f; check = c.p; break;
    case GMD_STREAM_SCALE: expect_val = s * 0.5f; check = b.p; break;
    case GMD_STREAM_ADD: expect_val = 3.f; check = c.p; break;
    case GMD_STREAM_TRIAD: expect_val = 2.f + s * 0.5f; check = a.p; break;
    case GMD_STREAM_WRITE: expect_val = s; check = a.p; break;
    default: break;
  }
  out->verified = 1;
  if (check) {
    for (int k = 0; k < 64 && out->verified; k++) {
      uint64_t idx = (n4 * 4 - 1) * (uint64_t)k / 63;
      float v = 0;
      GMD_HIP(hipMemcpy(&v, (const float*)check + idx, sizeof v, hipMemcpyDeviceToHost));
      if (v != expect_val) out->verified = 0;
    }
  } else {  // READ: the slot sums must add up to n (every element is 1.0; slot sums are exact integers)
    std::vector<float> part(GMD_STREAM_SINK_FLOATS);
    GMD_HIP(hipMemcpy(part.data(), sink.p, part.size() * sizeof(float), hipMemcpyDeviceToHost));
    double total = 0;
    for (float p : part) total += p;
    (void)grid;
    double want = (double)n4 * 4.0 * (iters + 1);  // the warm-up and every timed launch added into it
    if (total < want * 0.999 || total > want * 1.001) out->verified = 0;
  }
  std::vector<float> sorted = ms;
  std::sort(sorted.begin(), sorted.end());
  out->bytes_per_iter = bytes * (uint64_t)bytes_factor(kernel);
  out->ms_best = sorted.front();
  out->tbps_best = (double)out->bytes_per_iter / (sorted.front() * 1e-3) / 1e12;
  out->tbps_median = (double)out->bytes_per_iter / (sorted[sorted.size() / 2] * 1e-3) / 1e12;
  out->iters = iters;
  if (!out->verified) {
    gmd::set_error("stream kernel %d produced a wrong result", kernel);
    return GMD_ERR_VERIFY;
  }
  return GMD_OK;
}

}  // extern "C"
