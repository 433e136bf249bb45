// Does a vector-memory instruction issued with EXEC = 0 take part in the vmcnt queue?
// One wave: a load that misses every cache (a fresh 64 MiB stride), then ten loads with EXEC = 0, then
// s_waitcnt vmcnt(10).  If the masked loads are counted, the wait can only pass once the real load has returned
// (the counter retires in order): hundreds of cycles.  If they are not, one load is outstanding, 1 <= 10, and the
// wait passes at once.  A control run issues the ten loads with EXEC = -1 (must be slow) and a second control
// waits for vmcnt(11) (must be fast).
//   hipcc --offload-arch=gfx950 -O2 -o build/exec0_vmcnt_probe scripts/probes/exec0_vmcnt_probe.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

template <int MODE>  // 0: masked loads, wait 10;  1: unmasked loads, wait 10;  2: masked loads, wait 11
__global__ void probe(const float* __restrict__ far, const float* __restrict__ near, unsigned long long* out, int iter) {
  const float* p = far + ((size_t)MODE * 24 + iter) * (16u << 20) + threadIdx.x;  // a line nobody has read before
  const float* q = near + threadIdx.x;
  float v, sink[10];
  unsigned long long t0, t1;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
  asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
  if (MODE != 1) asm volatile("s_mov_b64 exec, 0" ::: "memory");
#pragma unroll
  for (int i = 0; i < 10; i++) asm volatile("global_load_dword %0, %1, off" : "=v"(sink[i]) : "v"(q) : "memory");
  asm volatile("s_mov_b64 exec, -1" ::: "memory");
  if (MODE == 2) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  float acc = v;
#pragma unroll
  for (int i = 0; i < 10; i++) acc += sink[i];
  if (threadIdx.x == 0) out[iter] = t1 - t0;
  if (acc == 12345.678f) out[iter] = 0;  // keep the loads alive
}

int main() {
  const int iters = 24;
  float *far_buf, *near_buf;
  unsigned long long* out;
  if (hipMalloc(&far_buf, (size_t)3 * iters * (64u << 20) + 4096) != hipSuccess) return 2;
  hipMalloc(&near_buf, 4096);
  hipMalloc(&out, sizeof(unsigned long long) * iters);
  hipMemset(far_buf, 0, (size_t)3 * iters * (64u << 20) + 4096);
  hipMemset(near_buf, 0, 4096);
  const char* names[3] = {"masked loads, vmcnt(10)", "unmasked loads, vmcnt(10)", "masked loads, vmcnt(11)"};
  const int order[3] = {1, 0, 2};
  for (int oi = 0; oi < 3; oi++) {
    const int mode = order[oi];
    std::vector<unsigned long long> h(iters);
    for (int it = 0; it < iters; it++) {
      // touch the near line first so that only the far load is slow
      if (mode == 0) probe<0><<<1, 64>>>(far_buf, near_buf, out, it);
      else if (mode == 1) probe<1><<<1, 64>>>(far_buf, near_buf, out, it);
      else probe<2><<<1, 64>>>(far_buf, near_buf, out, it);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 3;
    hipMemcpy(h.data(), out, sizeof(unsigned long long) * iters, hipMemcpyDeviceToHost);
    unsigned long long lo = ~0ull, hi = 0, sum = 0;
    for (int it = 4; it < iters; it++) {
      lo = h[it] < lo ? h[it] : lo;
      hi = h[it] > hi ? h[it] : hi;
      sum += h[it];
    }
    printf("%-28s ticks min %llu mean %llu max %llu\n", names[mode], lo, sum / (iters - 4), hi);
  }
  return 0;
}
