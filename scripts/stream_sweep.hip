// Tuning harness for the HBM stream kernels: sweeps grid size, unroll depth,
// traversal order and cache policy for copy / read / write on one MI355X and
// prints TB/s per variant.  Build: hipcc --offload-arch=gfx950 -O3 -o build/stream_sweep scripts/stream_sweep.hip
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x)                                                         \
  do {                                                                \
    hipError_t e = (x);                                               \
    if (e != hipSuccess) {                                            \
      printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); \
      return 1;                                                       \
    }                                                                 \
  } while (0)

template <bool NT>
__device__ __forceinline__ f32x4 ld(const f32x4* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p); else return *p;
}
template <bool NT>
__device__ __forceinline__ void st(f32x4* p, f32x4 v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// MODE 0: grid-stride, unrolled accesses one grid apart.
// MODE 1: each block owns a contiguous span; unrolled accesses one block (256 lanes * 16 B = 4 KiB) apart.
template <int OP, int U, int MODE, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ a, f32x4* __restrict__ c, uint64_t n4, float* sink) {
  f32x4 acc = {0, 0, 0, 0};
  if constexpr (MODE == 0) {
    const uint64_t stride = (uint64_t)gridDim.x * 256;
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
      f32x4 x[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if constexpr (OP != 2) x[u] = ld<NTL>(a + i + u * stride);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if constexpr (OP == 0) st<NTS>(c + i + u * stride, x[u]);
        if constexpr (OP == 1) acc += x[u];
        if constexpr (OP == 2) st<NTS>(c + i + u * stride, f32x4{1, 2, 3, 4});
      }
    }
  } else {
    const uint64_t per_block = n4 / gridDim.x;  // caller guarantees divisibility by 256*U
    uint64_t i = (uint64_t)blockIdx.x * per_block + threadIdx.x;
    const uint64_t end = (uint64_t)(blockIdx.x + 1) * per_block;
    for (; i < end; i += 256 * U) {
      f32x4 x[U];
#pragma unroll
      for (int u = 0; u < U; u++)
        if constexpr (OP != 2) x[u] = ld<NTL>(a + i + u * 256);
#pragma unroll
      for (int u = 0; u < U; u++) {
        if constexpr (OP == 0) st<NTS>(c + i + u * 256, x[u]);
        if constexpr (OP == 1) acc += x[u];
        if constexpr (OP == 2) st<NTS>(c + i + u * 256, f32x4{1, 2, 3, 4});
      }
    }
  }
  if constexpr (OP == 1)
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

template <int OP, int U, int MODE, bool NTL, bool NTS>
double run(const f32x4* a, f32x4* c, uint64_t n4, int blocks, float* sink, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<OP, U, MODE, NTL, NTS><<<blocks, 256>>>(a, c, n4, sink);
  hipDeviceSynchronize();
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; i++) {
    hipEventRecord(e0);
    k<OP, U, MODE, NTL, NTS><<<blocks, 256>>>(a, c, n4, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms[i], e0, e1);
  }
  std::sort(ms.begin(), ms.end());
  double bytes = (double)n4 * 16 * (OP == 0 ? 2 : 1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  return bytes / (ms[iters / 2] * 1e-3) / 1e12;
}

// MODE 2: one access per lane and nothing else: workgroup w touches elements [w*BS, (w+1)*BS).  The dispatcher hands
// workgroups out in order, so the addresses in flight are one compact window that slides through the buffers.
template <int OP, int BS, bool NTL, bool NTS>
__global__ __launch_bounds__(BS) void k1(const f32x4* __restrict__ a, const f32x4* __restrict__ b, f32x4* __restrict__ c, uint64_t n4) {
  extern __shared__ char occupancy_limiter[];
  const uint64_t i = (uint64_t)blockIdx.x * BS + threadIdx.x;
  if (i >= n4) return;
  if constexpr (OP == 0) st<NTS>(c + i, ld<NTL>(a + i));
  if constexpr (OP == 2) st<NTS>(c + i, f32x4{1, 2, 3, 4});
  if constexpr (OP == 3) st<NTS>(c + i, ld<NTL>(a + i) + 3.0f * ld<NTL>(b + i));
}

template <int OP, int BS, bool NTL, bool NTS>
double run1(const f32x4* a, const f32x4* b, f32x4* c, uint64_t n4, int lds, int iters) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const unsigned grid = (unsigned)((n4 + BS - 1) / BS);
  if (lds > 48 * 1024) (void)hipFuncSetAttribute((const void*)k1<OP, BS, NTL, NTS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  k1<OP, BS, NTL, NTS><<<grid, BS, lds>>>(a, b, c, n4);
  (void)hipDeviceSynchronize();
  std::vector<float> ms(iters);
  for (int i = 0; i < iters; i++) {
    (void)hipEventRecord(e0);
    k1<OP, BS, NTL, NTS><<<grid, BS, lds>>>(a, b, c, n4);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms[i], e0, e1);
  }
  std::sort(ms.begin(), ms.end());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return (double)n4 * 16 * (OP == 0 ? 2 : OP == 3 ? 3 : 1) / (ms[iters / 2] * 1e-3) / 1e12;
}

template <int BS>
void sweep1(const f32x4* a, const f32x4* b, f32x4* c, uint64_t n4) {
  for (int lds : {0, 20 * 1024, 40 * 1024, 80 * 1024}) {
    printf("onepass block=%-4d lds=%-5d copy plain=%.3f nt=%.3f ntload=%.3f | write plain=%.3f nt=%.3f | triad plain=%.3f nt=%.3f ntload=%.3f TB/s\n",
           BS, lds, run1<0, BS, false, false>(a, b, c, n4, lds, 9), run1<0, BS, true, true>(a, b, c, n4, lds, 9),
           run1<0, BS, true, false>(a, b, c, n4, lds, 9), run1<2, BS, false, false>(a, b, c, n4, lds, 9),
           run1<2, BS, false, true>(a, b, c, n4, lds, 9), run1<3, BS, false, false>(a, b, c, n4, lds, 9),
           run1<3, BS, true, true>(a, b, c, n4, lds, 9), run1<3, BS, true, false>(a, b, c, n4, lds, 9));
    fflush(stdout);
  }
}

int g_min_bpc = 2;  // argv[2]: smallest blocks/CU to try

template <int OP, int U, int MODE>
void sweep_nt(const char* name, const f32x4* a, f32x4* c, uint64_t n4, float* sink) {
  // Up to one pass per workgroup (4 GiB: 4096 blocks/CU at U=1 is a 4 KiB span, a million workgroups).
  const int bpc[] = {2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 2048, 4096};
  for (int b : bpc) {
    if (b < g_min_bpc) continue;
    int blocks = 256 * b;
    if (MODE == 1 && (n4 % ((uint64_t)blocks * 256 * U) || n4 / blocks < 256u * U)) continue;
    double p = run<OP, U, MODE, false, false>(a, c, n4, blocks, sink, 7);
    double q = run<OP, U, MODE, true, true>(a, c, n4, blocks, sink, 7);
    double r = OP == 0 ? run<OP, U, MODE, true, false>(a, c, n4, blocks, sink, 7) : 0;
    double s = OP == 0 ? run<OP, U, MODE, false, true>(a, c, n4, blocks, sink, 7) : 0;
    printf("%-5s mode=%d unroll=%d blocks/CU=%-2d plain=%.3f nt=%.3f ntload=%.3f ntstore=%.3f TB/s\n", name, MODE, U, b, p, q, r, s);
    fflush(stdout);
  }
}

template <int OP>
void sweep(const char* name, const f32x4* a, f32x4* c, uint64_t n4, float* sink) {
  if (g_min_bpc < 128) {  // the grid-stride arm never wins; skip it in the fine-span sweep
    sweep_nt<OP, 1, 0>(name, a, c, n4, sink);
    sweep_nt<OP, 2, 0>(name, a, c, n4, sink);
    sweep_nt<OP, 4, 0>(name, a, c, n4, sink);
    sweep_nt<OP, 8, 0>(name, a, c, n4, sink);
  }
  sweep_nt<OP, 1, 1>(name, a, c, n4, sink);
  sweep_nt<OP, 2, 1>(name, a, c, n4, sink);
  sweep_nt<OP, 4, 1>(name, a, c, n4, sink);
  sweep_nt<OP, 8, 1>(name, a, c, n4, sink);
  sweep_nt<OP, 16, 1>(name, a, c, n4, sink);
}

int main(int argc, char** argv) {
  uint64_t gib = argc > 1 ? atoll(argv[1]) : 4;
  if (argc > 2) g_min_bpc = atoi(argv[2]);
  uint64_t bytes = gib << 30, n4 = bytes / 16;
  f32x4 *a, *c;
  float* sink;
  CK(hipMalloc(&a, bytes));
  CK(hipMalloc(&c, bytes));
  CK(hipMalloc(&sink, 4096));
  CK(hipMemset(a, 1, bytes));
  CK(hipMemset(c, 0, bytes));
  CK(hipDeviceSynchronize());
  if (g_min_bpc == 0) {  // the one-pass arm only
    f32x4* b;
    CK(hipMalloc(&b, bytes));
    CK(hipMemset(b, 1, bytes));
    CK(hipDeviceSynchronize());
    sweep1<64>(a, b, c, n4);
    sweep1<128>(a, b, c, n4);
    sweep1<256>(a, b, c, n4);
    sweep1<512>(a, b, c, n4);
    sweep1<1024>(a, b, c, n4);
    return 0;
  }
  // hipMemcpy D2D as the runtime's own reference
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int i = 0; i < 3; i++) {
    hipEventRecord(e0);
    CK(hipMemcpyAsync(c, a, bytes, hipMemcpyDeviceToDevice, 0));
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyD2D %.3f TB/s\n", 2.0 * bytes / (ms * 1e-3) / 1e12);
  }
  sweep<0>("copy", a, c, n4, sink);
  sweep<1>("read", a, c, n4, sink);
  sweep<2>("write", a, c, n4, sink);
  return 0;
}
