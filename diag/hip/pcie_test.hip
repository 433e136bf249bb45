// Host link test: what the PCIe path between host memory and one GPU delivers, and whether it delivers the data.
//
// The reference prices a PCIe link from its generation and width (bindings/go/nvml/nvml.go:314-326: MB/s per lane
// times lanes) and reads replay counters; it never moves a byte.  Here the link is loaded five ways over pinned host
// memory: the copy engines host-to-device, device-to-host and both at once (what checkpoint loads and activation
// offload do), and a kernel that reads from, respectively writes to, host memory directly with 16 bytes per lane
// (what zero-copy data loaders and host-resident tables do: the shader's own path over the link, no copy engine).
// Every transfer carries a position-dependent pattern and is verified on the far side, so a link that corrupts
// fails even when it is fast.  Two 8-byte copies give the latency floor of a transfer.
#include <chrono>

#include "diag_common.h"

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__host__ __device__ inline u32x4 host_pattern(uint64_t w, uint32_t seed) {
  const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
  u32x4 v;
  v.x = lo * 2654435761u + seed;
  v.y = (lo ^ 0x9e3779b9u) + hi;
  v.z = ~lo + (seed << 24);
  v.w = lo + 0x7f4a7c15u * (seed + 1);
  return v;
}

__global__ void pcie_fill_kernel(u32x4* __restrict__ p, uint64_t words, uint32_t seed) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += stride) __builtin_nontemporal_store(host_pattern(w, seed), p + w);
}

// Reads `p` (device or host memory) against the pattern; four independent 16-byte loads in flight per lane, since
// over the link it is the number of outstanding reads that makes the rate.
__global__ __launch_bounds__(256) void pcie_check_kernel(const u32x4* __restrict__ p, uint64_t words, uint32_t seed, unsigned long long* bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  auto differs = [&](const u32x4& got, uint64_t at) {
    const u32x4 want = host_pattern(at, seed);
    return got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w;
  };
  for (; w + 3 * stride < words; w += 4 * stride) {
    const u32x4 a = __builtin_nontemporal_load(p + w), b = __builtin_nontemporal_load(p + w + stride);
    const u32x4 c = __builtin_nontemporal_load(p + w + 2 * stride), d = __builtin_nontemporal_load(p + w + 3 * stride);
    mine += differs(a, w) + differs(b, w + stride) + differs(c, w + 2 * stride) + differs(d, w + 3 * stride);
  }
  for (; w < words; w += stride) mine += differs(__builtin_nontemporal_load(p + w), w);
  if (mine) atomicAdd(bad, mine);
}

uint64_t host_check(const u32x4* p, uint64_t words, uint32_t seed) {
  uint64_t bad = 0;
  for (uint64_t w = 0; w < words; w++) {
    const u32x4 want = host_pattern(w, seed), got = p[w];
    bad += got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w;
  }
  return bad;
}

struct HostBuffer {
  void* p = nullptr;
  ~HostBuffer() {
    if (p) (void)hipHostFree(p);
  }
};
struct Ev {
  hipEvent_t e = nullptr;
  ~Ev() {
    if (e) (void)hipEventDestroy(e);
  }
};
struct St {
  hipStream_t s = nullptr;
  ~St() {
    if (s) (void)hipStreamDestroy(s);
  }
};

}  // namespace

extern "C" int gmd_pcie_test(int device, const gmd_pcie_config_t* cfg_in, gmd_pcie_result_t* out) {
  if (!out) return GMD_ERR_BADPARAM;
  gmd_pcie_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  if (cfg_in) cfg = *cfg_in;
  memset(out, 0, sizeof *out);
  int visible = 0;
  GMD_HIP(hipGetDeviceCount(&visible));
  if (device < 0 || device >= visible) {
    gmd::set_error("device %d of %d", device, visible);
    return GMD_ERR_NO_DEVICE;
  }
  GMD_HIP(hipSetDevice(device));
  const uint64_t bytes = std::min<uint64_t>(cfg.bytes ? cfg.bytes : 256ull << 20, 8ull << 30) & ~255ull;
  if (bytes < 4096) {
    gmd::set_error("%llu bytes per transfer: at least 4096", (unsigned long long)bytes);
    return GMD_ERR_BADPARAM;
  }
  const uint64_t words = bytes / 16;
  const int iters = cfg.iters > 0 ? cfg.iters : 5;
  const bool verify = cfg.verify != 0;
  const unsigned grid = cfg.workgroups > 0 ? (unsigned)std::min(cfg.workgroups, 4096) : 1024;  // 4 per CU: reads in flight, not ALU, make the rate

  HostBuffer up, down;  // host -> device source, device -> host destination
  GMD_HIP(hipHostMalloc(&up.p, bytes, hipHostMallocDefault));
  GMD_HIP(hipHostMalloc(&down.p, bytes, hipHostMallocDefault));
  gmd::DeviceBuffer dev_in, dev_out, d_bad;
  int rc;
  if ((rc = dev_in.alloc(bytes)) || (rc = dev_out.alloc(bytes)) || (rc = d_bad.alloc(sizeof(unsigned long long)))) return rc;
  void* up_dev = nullptr;
  void* down_dev = nullptr;  // the host buffers as the GPU addresses them
  GMD_HIP(hipHostGetDevicePointer(&up_dev, up.p, 0));
  GMD_HIP(hipHostGetDevicePointer(&down_dev, down.p, 0));
  St s0, s1;
  GMD_HIP(hipStreamCreateWithFlags(&s0.s, hipStreamNonBlocking));
  GMD_HIP(hipStreamCreateWithFlags(&s1.s, hipStreamNonBlocking));
  Ev a0, a1, b0, b1;
  GMD_HIP(hipEventCreate(&a0.e));
  GMD_HIP(hipEventCreate(&a1.e));
  GMD_HIP(hipEventCreate(&b0.e));
  GMD_HIP(hipEventCreate(&b1.e));

  const uint32_t seed_up = 0x51 + (uint32_t)device, seed_down = 0xa7 + (uint32_t)device, seed_kw = 0x3c + (uint32_t)device;
  {
    u32x4* h = (u32x4*)up.p;
    for (uint64_t w = 0; w < words; w++) h[w] = host_pattern(w, seed_up);
  }
  memset(down.p, 0, bytes);
  pcie_fill_kernel<<<gmd::stream_grid(words, 256, 4), 256, 0, s0.s>>>((u32x4*)dev_out.p, words, seed_down);
  GMD_HIP(hipGetLastError());
  GMD_HIP(hipMemsetAsync(dev_in.p, 0, bytes, s0.s));
  GMD_HIP(hipStreamSynchronize(s0.s));

  auto gbps_of = [&](float ms) { return ms > 0 ? (double)bytes / (ms * 1e-3) / 1e9 : 0.0; };
  auto device_bad = [&](const void* p, uint32_t seed, uint64_t* n) -> int {
    GMD_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(unsigned long long), s0.s));
    pcie_check_kernel<<<grid, 256, 0, s0.s>>>((const u32x4*)p, words, seed, (unsigned long long*)d_bad.p);
    GMD_HIP(hipGetLastError());
    unsigned long long v = 0;
    GMD_HIP(hipMemcpyAsync(&v, d_bad.p, sizeof v, hipMemcpyDeviceToHost, s0.s));
    GMD_HIP(hipStreamSynchronize(s0.s));
    *n = v;
    return GMD_OK;
  };

  // 1. copy engines, one direction at a time
  for (int it = 0; it < iters + 1; it++) {  // the first transfer maps the pages: not counted
    float ms = 0;
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    GMD_HIP(hipMemcpyAsync(dev_in.p, up.p, bytes, hipMemcpyHostToDevice, s0.s));
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventElapsedTime(&ms, a0.e, a1.e));
    if (it) out->h2d_gbps = std::max(out->h2d_gbps, gbps_of(ms));
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    GMD_HIP(hipMemcpyAsync(down.p, dev_out.p, bytes, hipMemcpyDeviceToHost, s0.s));
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventElapsedTime(&ms, a0.e, a1.e));
    if (it) out->d2h_gbps = std::max(out->d2h_gbps, gbps_of(ms));
  }
  if (verify) {
    uint64_t n = 0;
    if ((rc = device_bad(dev_in.p, seed_up, &n))) return rc;
    out->bad_h2d = n;
    out->bad_d2h = host_check((const u32x4*)down.p, words, seed_down);
  }

  // 2. both directions at once, each on a stream of its own
  for (int it = 0; it < iters; it++) {
    float ms0 = 0, ms1 = 0;
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    GMD_HIP(hipEventRecord(b0.e, s1.s));
    GMD_HIP(hipMemcpyAsync(dev_in.p, up.p, bytes, hipMemcpyHostToDevice, s0.s));
    GMD_HIP(hipMemcpyAsync(down.p, dev_out.p, bytes, hipMemcpyDeviceToHost, s1.s));
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventRecord(b1.e, s1.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventSynchronize(b1.e));
    GMD_HIP(hipEventElapsedTime(&ms0, a0.e, a1.e));
    GMD_HIP(hipEventElapsedTime(&ms1, b0.e, b1.e));
    // a pair counts only as a pair: the iteration with the largest sum
    if (gbps_of(ms0) + gbps_of(ms1) > out->bidir_h2d_gbps + out->bidir_d2h_gbps) out->bidir_h2d_gbps = gbps_of(ms0), out->bidir_d2h_gbps = gbps_of(ms1);
  }

  // 3. the shader's own path: a kernel reads host memory (and checks it), a kernel writes host memory
  for (int it = 0; it < iters + 1; it++) {
    float ms = 0;
    GMD_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(unsigned long long), s0.s));
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    pcie_check_kernel<<<grid, 256, 0, s0.s>>>((const u32x4*)up_dev, words, seed_up, (unsigned long long*)d_bad.p);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventElapsedTime(&ms, a0.e, a1.e));
    if (it) out->kernel_read_gbps = std::max(out->kernel_read_gbps, gbps_of(ms));
    unsigned long long v = 0;
    GMD_HIP(hipMemcpy(&v, d_bad.p, sizeof v, hipMemcpyDeviceToHost));
    out->bad_kernel_read += v;
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    pcie_fill_kernel<<<grid, 256, 0, s0.s>>>((u32x4*)down_dev, words, seed_kw);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventElapsedTime(&ms, a0.e, a1.e));
    if (it) out->kernel_write_gbps = std::max(out->kernel_write_gbps, gbps_of(ms));
  }
  if (verify) out->bad_kernel_write = host_check((const u32x4*)down.p, words, seed_kw);

  // 3b. the link in both directions at once, driven by the two kernels on two streams: whether two copy-engine
  //     transfers overlap is the runtime's choice of engines (on some hosts they run one after the other); two
  //     kernels that fit on the chip together always do, so this is the link's duplex rate and not a scheduling fact
  for (int it = 0; it < iters + 1; it++) {
    float ms0 = 0, ms1 = 0;
    GMD_HIP(hipMemsetAsync(d_bad.p, 0, sizeof(unsigned long long), s0.s));
    GMD_HIP(hipStreamSynchronize(s0.s));
    GMD_HIP(hipEventRecord(a0.e, s0.s));
    GMD_HIP(hipEventRecord(b0.e, s1.s));
    pcie_check_kernel<<<grid / 2, 256, 0, s0.s>>>((const u32x4*)up_dev, words, seed_up, (unsigned long long*)d_bad.p);
    pcie_fill_kernel<<<grid / 2, 256, 0, s1.s>>>((u32x4*)down_dev, words, seed_kw);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipEventRecord(a1.e, s0.s));
    GMD_HIP(hipEventRecord(b1.e, s1.s));
    GMD_HIP(hipEventSynchronize(a1.e));
    GMD_HIP(hipEventSynchronize(b1.e));
    GMD_HIP(hipEventElapsedTime(&ms0, a0.e, a1.e));
    GMD_HIP(hipEventElapsedTime(&ms1, b0.e, b1.e));
    if (it && gbps_of(ms0) + gbps_of(ms1) > out->duplex_read_gbps + out->duplex_write_gbps)
      out->duplex_read_gbps = gbps_of(ms0), out->duplex_write_gbps = gbps_of(ms1);
    unsigned long long v = 0;
    GMD_HIP(hipMemcpy(&v, d_bad.p, sizeof v, hipMemcpyDeviceToHost));
    out->bad_kernel_read += v;
  }
  if (verify) out->bad_kernel_write += host_check((const u32x4*)down.p, words, seed_kw);

  // 4. eight bytes each way, host clock around a blocking copy: what one small transfer costs
  {
    uint64_t word = 0;
    double best_up = 1e30, best_down = 1e30;
    for (int it = 0; it < 64; it++) {
      auto t0 = std::chrono::steady_clock::now();
      GMD_HIP(hipMemcpy(dev_in.p, &word, sizeof word, hipMemcpyHostToDevice));
      auto t1 = std::chrono::steady_clock::now();
      GMD_HIP(hipMemcpy(&word, dev_in.p, sizeof word, hipMemcpyDeviceToHost));
      auto t2 = std::chrono::steady_clock::now();
      if (it >= 8) {
        best_up = std::min(best_up, std::chrono::duration<double, std::micro>(t1 - t0).count());
        best_down = std::min(best_down, std::chrono::duration<double, std::micro>(t2 - t1).count());
      }
    }
    out->small_h2d_us = best_up;
    out->small_d2h_us = best_down;
  }
  out->bytes = bytes;
  out->iters = iters;
  out->verified = verify ? 1 : 0;
  GMD_HIP(hipDeviceSynchronize());
  return GMD_OK;
}

// MB/s of one lane in one direction by generation: the reference's table (bindings/go/nvml/nvml.go:314-326 stops at
// generation 4), continued with the two generations since.
extern "C" double gmd_pcie_link_gbps(int gen, int width) {
  static const double lane_mbps[] = {0, 250, 500, 985, 1969, 3938, 7563};
  if (gen < 1 || gen > 6 || width < 1) return 0;
  return lane_mbps[gen] * width / 1000.0;
}
