// xGMI link test: a copy kernel that stores straight into a peer GPU's memory, and the grading of the
// resulting bandwidth matrix.
//
// The eight GPUs of an MI355X node are a full mesh of point-to-point links, so every ordered pair of GPUs is
// exactly one direction of one link.  A copy engine transfer (hipMemcpyPeer) measures the engine as much as the
// link; here the compute units of the source GPU read local HBM and store over the link, 16 bytes per lane with
// streaming stores, which is also what RCCL's kernels do.  Timing is by HIP events on the source GPU's stream.
//
// Three patterns tell apart what a single matrix cannot: one pair at a time (a slow lane), both directions of a
// pair at once (a link that only holds its rate half duplex), every GPU to every peer at once (links that
// share something they should not, or a GPU whose fabric port is the limit).
//
// Reference: the reference stack has no active link test; its topology sample only lists link types
// (bindings/go/samples/dcgm/topology/main.go:28-49).
#include <cmath>
#include <cstdint>

#include "diag_common.h"

namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// What word w of the buffer sent by device-list position `src` holds.
__device__ __forceinline__ u32x4 pattern_word(uint64_t w, uint32_t src) {
  const uint32_t lo = (uint32_t)w, hi = (uint32_t)(w >> 32);
  u32x4 v;
  v.x = lo * 2654435761u + src;
  v.y = (lo ^ 0x9e3779b9u) + hi;
  v.z = ~lo + (src << 24);
  v.w = lo + 0x7f4a7c15u * (src + 1);
  return v;
}

__global__ void link_fill_kernel(u32x4* p, uint64_t words, uint32_t src) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += stride) p[w] = pattern_word(w, src);
}

// local -> peer.  Each workgroup walks one contiguous span so that the stores of a wave are 1 KiB runs and the
// link sees long bursts; four independent 16-byte loads are in flight per lane before the first store.
__global__ __launch_bounds__(256) void link_copy_kernel(const u32x4* __restrict__ local, u32x4* __restrict__ peer, uint64_t words) {
  const uint64_t per = (words + gridDim.x - 1) / gridDim.x;
  const uint64_t begin = (uint64_t)blockIdx.x * per;
  const uint64_t end = begin + per < words ? begin + per : words;
  uint64_t w = begin + threadIdx.x;
  for (; w + 3 * 256 < end; w += 4 * 256) {
    u32x4 a = __builtin_nontemporal_load(local + w);
    u32x4 b = __builtin_nontemporal_load(local + w + 256);
    u32x4 c = __builtin_nontemporal_load(local + w + 512);
    u32x4 d = __builtin_nontemporal_load(local + w + 768);
    __builtin_nontemporal_store(a, peer + w);
    __builtin_nontemporal_store(b, peer + w + 256);
    __builtin_nontemporal_store(c, peer + w + 512);
    __builtin_nontemporal_store(d, peer + w + 768);
  }
  for (; w < end; w += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(local + w), peer + w);
}

__global__ void link_check_kernel(const u32x4* p, uint64_t words, uint32_t src, unsigned long long* bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  for (uint64_t w = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; w < words; w += stride) {
    const u32x4 got = p[w], want = pattern_word(w, src);
    if (got.x != want.x || got.y != want.y || got.z != want.z || got.w != want.w) mine++;
  }
  if (mine) atomicAdd(bad, mine);
}

struct Peer {  // one ordered pair
  hipStream_t stream = nullptr;
  hipEvent_t start = nullptr, stop = nullptr;
  bool ok = false;
};

struct LinkRig {
  std::vector<int> ids;
  uint64_t words = 0;
  int iters = 5, grid = 64;
  std::vector<void*> src;               // per GPU: the buffer it sends
  std::vector<void*> dst;               // per GPU: ndev slots of `words`, slot s receives from position s
  std::vector<unsigned long long*> bad; // per GPU: one counter
  std::vector<Peer> pairs;              // [i*n+j]

  ~LinkRig() {
    const int n = (int)ids.size();
    for (int i = 0; i < n; i++) {
      (void)hipSetDevice(ids[i]);
      for (int j = 0; j < n && !pairs.empty(); j++) {
        Peer& p = pairs[(size_t)i * n + j];
        if (p.start) (void)hipEventDestroy(p.start);
        if (p.stop) (void)hipEventDestroy(p.stop);
        if (p.stream) (void)hipStreamDestroy(p.stream);
      }
      if (i < (int)src.size() && src[i]) (void)hipFree(src[i]);
      if (i < (int)dst.size() && dst[i]) (void)hipFree(dst[i]);
      if (i < (int)bad.size() && bad[i]) (void)hipFree(bad[i]);
    }
  }

  u32x4* slot(int dst_pos, int src_pos) const { return (u32x4*)dst[dst_pos] + (uint64_t)src_pos * words; }

  int setup() {
    const int n = (int)ids.size();
    src.assign(n, nullptr), dst.assign(n, nullptr), bad.assign(n, nullptr);
    pairs.assign((size_t)n * n, Peer());
    for (int i = 0; i < n; i++) {
      GMD_HIP(hipSetDevice(ids[i]));
      GMD_HIP(hipMalloc(&src[i], words * 16));
      GMD_HIP(hipMalloc(&dst[i], words * 16 * n));
      GMD_HIP(hipMalloc((void**)&bad[i], sizeof(unsigned long long)));
      link_fill_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>((u32x4*)src[i], words, (uint32_t)i);
      GMD_HIP(hipGetLastError());
      GMD_HIP(hipMemset(dst[i], 0, words * 16 * n));
      for (int j = 0; j < n; j++) {
        if (i == j) continue;
        Peer& p = pairs[(size_t)i * n + j];
        int can = 0;
        GMD_HIP(hipDeviceCanAccessPeer(&can, ids[i], ids[j]));
        if (!can) continue;
        hipError_t e = hipDeviceEnablePeerAccess(ids[j], 0);
        (void)hipGetLastError();
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) continue;
        GMD_HIP(hipStreamCreateWithFlags(&p.stream, hipStreamNonBlocking));
        GMD_HIP(hipEventCreate(&p.start));
        GMD_HIP(hipEventCreate(&p.stop));
        p.ok = true;
      }
      GMD_HIP(hipDeviceSynchronize());
    }
    return GMD_OK;
  }

  // Queues warm-up + timed transfers for i -> j on the pair's stream (asynchronous).
  int enqueue(int i, int j) {
    const int n = (int)ids.size();
    Peer& p = pairs[(size_t)i * n + j];
    GMD_HIP(hipSetDevice(ids[i]));
    link_copy_kernel<<<grid, 256, 0, p.stream>>>((const u32x4*)src[i], slot(j, i), words);
    GMD_HIP(hipEventRecord(p.start, p.stream));
    for (int k = 0; k < iters; k++) link_copy_kernel<<<grid, 256, 0, p.stream>>>((const u32x4*)src[i], slot(j, i), words);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipEventRecord(p.stop, p.stream));
    return GMD_OK;
  }

  int collect(int i, int j, double* gbps) {
    const int n = (int)ids.size();
    Peer& p = pairs[(size_t)i * n + j];
    GMD_HIP(hipSetDevice(ids[i]));
    GMD_HIP(hipEventSynchronize(p.stop));
    float ms = 0;
    GMD_HIP(hipEventElapsedTime(&ms, p.start, p.stop));
    *gbps = ms > 0 ? (double)words * 16 * iters / (ms * 1e-3) / 1e9 : 0;
    return GMD_OK;
  }

  // Data check on the destination GPU, after the source's stream has drained.
  int check(int i, int j, uint64_t* out) {
    GMD_HIP(hipSetDevice(ids[j]));
    GMD_HIP(hipMemset(bad[j], 0, sizeof(unsigned long long)));
    link_check_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>(slot(j, i), words, (uint32_t)i, bad[j]);
    GMD_HIP(hipGetLastError());
    unsigned long long v = 0;
    GMD_HIP(hipMemcpy(&v, bad[j], sizeof v, hipMemcpyDeviceToHost));
    *out = v;
    // the next pattern must not pass on this pattern's data
    GMD_HIP(hipMemset(slot(j, i), 0, words * 16));
    return GMD_OK;
  }
};

double median_of(std::vector<double> v) {
  if (v.empty()) return 0;
  std::sort(v.begin(), v.end());
  const size_t m = v.size() / 2;
  return v.size() % 2 ? v[m] : 0.5 * (v[m - 1] + v[m]);
}

}  // namespace

extern "C" {

int gmd_link_test(const gmd_link_config_t* cfg_in, int mode, double* gbps, uint64_t* bad) {
  if (!gbps || mode < GMD_LINK_UNI || mode > GMD_LINK_ALL) return GMD_ERR_BADPARAM;
  gmd_link_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  if (cfg_in) cfg = *cfg_in;
  int visible = 0;
  GMD_HIP(hipGetDeviceCount(&visible));
  int n = cfg.ndev > 0 ? cfg.ndev : visible;
  if (n <= 0 || n > visible || n > 64) {
    gmd::set_error("%d GPUs requested, %d visible", n, visible);
    return GMD_ERR_BADPARAM;
  }
  LinkRig rig;
  for (int i = 0; i < n; i++) {
    const int id = cfg.devices ? cfg.devices[i] : i;
    if (id < 0 || id >= visible || std::find(rig.ids.begin(), rig.ids.end(), id) != rig.ids.end()) {
      gmd::set_error("bad or repeated device index %d", id);
      return GMD_ERR_BADPARAM;
    }
    rig.ids.push_back(id);
  }
  for (int i = 0; i < n * n; i++) {
    gbps[i] = 0;
    if (bad) bad[i] = 0;
  }
  if (n == 1) return GMD_OK;  // no link to test

  const uint64_t bytes = cfg.bytes ? cfg.bytes : 256ull << 20;
  rig.words = std::max<uint64_t>(256, bytes / 16);
  rig.iters = cfg.iters > 0 ? cfg.iters : 5;
  rig.grid = cfg.workgroups > 0 ? std::min(cfg.workgroups, 4096) : 64;
  int rc = rig.setup();
  if (rc) return rc;
  auto ok = [&](int i, int j) { return i != j && rig.pairs[(size_t)i * n + j].ok; };
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (i != j && !ok(i, j)) gbps[i * n + j] = -1;

  auto finish = [&](int i, int j) -> int {
    int r = rig.collect(i, j, &gbps[i * n + j]);
    if (r) return r;
    uint64_t b = 0;
    if (cfg.verify && (r = rig.check(i, j, &b))) return r;
    if (bad) bad[i * n + j] = b;
    return GMD_OK;
  };

  if (mode == GMD_LINK_UNI) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (ok(i, j) && ((rc = rig.enqueue(i, j)) || (rc = finish(i, j)))) return rc;
  } else if (mode == GMD_LINK_BIDI) {
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) {
        if (!ok(i, j) || !ok(j, i)) continue;
        if ((rc = rig.enqueue(i, j)) || (rc = rig.enqueue(j, i))) return rc;
        if ((rc = finish(i, j)) || (rc = finish(j, i))) return rc;
      }
  } else {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (ok(i, j) && (rc = rig.enqueue(i, j))) return rc;
    // every stream drains before the first check clears a slot
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (ok(i, j) && (rc = rig.collect(i, j, &gbps[i * n + j]))) return rc;
    if (cfg.verify)
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          uint64_t b = 0;
          if (ok(i, j) && (rc = rig.check(i, j, &b))) return rc;
          if (bad) bad[i * n + j] = b;
        }
  }
  return GMD_OK;
}

int gmd_link_selftest(int device, uint64_t bytes, int workgroups, double* gbps, uint64_t* bad) {
  if (!gbps || !bad) return GMD_ERR_BADPARAM;
  const uint64_t words = std::max<uint64_t>(1, (bytes ? bytes : 256ull << 20) / 16);
  const int grid = workgroups > 0 ? std::min(workgroups, 4096) : 64;
  GMD_HIP(hipSetDevice(device));
  gmd::DeviceBuffer src, dst, cnt;
  int rc;
  // one guard word past the destination: the kernel must leave it alone
  if ((rc = src.alloc(words * 16)) || (rc = dst.alloc((words + 1) * 16)) || (rc = cnt.alloc(sizeof(unsigned long long)))) return rc;
  link_fill_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>((u32x4*)src.p, words, 5u);
  GMD_HIP(hipMemset(dst.p, 0xa5, (words + 1) * 16));
  GMD_HIP(hipMemset(cnt.p, 0, sizeof(unsigned long long)));
  hipEvent_t e0, e1;
  GMD_HIP(hipEventCreate(&e0));
  GMD_HIP(hipEventCreate(&e1));
  link_copy_kernel<<<grid, 256>>>((const u32x4*)src.p, (u32x4*)dst.p, words);
  GMD_HIP(hipEventRecord(e0, nullptr));
  for (int k = 0; k < 5; k++) link_copy_kernel<<<grid, 256>>>((const u32x4*)src.p, (u32x4*)dst.p, words);
  GMD_HIP(hipEventRecord(e1, nullptr));
  GMD_HIP(hipGetLastError());
  GMD_HIP(hipEventSynchronize(e1));
  float ms = 0;
  GMD_HIP(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  *gbps = ms > 0 ? (double)words * 16 * 5 / (ms * 1e-3) / 1e9 : 0;
  link_check_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>((const u32x4*)dst.p, words, 5u, (unsigned long long*)cnt.p);
  GMD_HIP(hipGetLastError());
  unsigned long long v = 0;
  uint32_t guard[4];
  GMD_HIP(hipMemcpy(&v, cnt.p, sizeof v, hipMemcpyDeviceToHost));
  GMD_HIP(hipMemcpy(guard, (char*)dst.p + words * 16, 16, hipMemcpyDeviceToHost));
  for (uint32_t g : guard)
    if (g != 0xa5a5a5a5u) v++;
  *bad = v;
  return GMD_OK;
}

int gmd_link_grade(int ndev, const double* gbps, const uint64_t* bad, const gmd_link_rule_t* rule_in,
                   gmd_link_verdict_t* out, int* n_out) {
  if (ndev < 1 || !gbps || !out || !n_out) return GMD_ERR_BADPARAM;
  gmd_link_rule_t rule = {0, 0, 0};
  if (rule_in) rule = *rule_in;
  if (rule.warn_ratio <= 0) rule.warn_ratio = 0.90;
  if (rule.fail_ratio <= 0) rule.fail_ratio = 0.75;
  if (rule.fail_ratio > rule.warn_ratio) rule.fail_ratio = rule.warn_ratio;
  const int cap = *n_out;
  *n_out = 0;
  int worst = 0;
  for (int i = 0; i < ndev; i++)
    for (int j = 0; j < ndev; j++) {
      if (i == j) continue;
      if (*n_out >= cap) return GMD_ERR_BADPARAM;
      gmd_link_verdict_t v;
      memset(&v, 0, sizeof v);
      v.src = i, v.dst = j;
      v.gbps = gbps[i * ndev + j];
      std::vector<double> others;
      for (int a = 0; a < ndev; a++)
        for (int b = 0; b < ndev; b++)
          if (a != b && !(a == i && b == j) && gbps[a * ndev + b] > 0 && std::isfinite(gbps[a * ndev + b]))
            others.push_back(gbps[a * ndev + b]);
      const double med = others.size() >= 2 ? median_of(others) : 0;
      v.sibling_median_gbps = med;
      const uint64_t nbad = bad ? bad[i * ndev + j] : 0;
      if (v.gbps < 0 || !std::isfinite(v.gbps)) {
        v.health = 20;
        snprintf(v.message, sizeof v.message, "no peer access");
      } else if (nbad) {
        v.health = 20;
        snprintf(v.message, sizeof v.message, "%llu corrupt 16-byte words", (unsigned long long)nbad);
      } else if (rule.floor_gbps > 0 && v.gbps < rule.floor_gbps) {
        v.health = 20;
        snprintf(v.message, sizeof v.message, "%.1f GB/s is under the floor of %.1f GB/s", v.gbps, rule.floor_gbps);
      } else if (med > 0 && v.gbps < rule.fail_ratio * med) {
        v.health = 20;
        snprintf(v.message, sizeof v.message, "%.1f GB/s is %.0f %% of the other links' median %.1f GB/s", v.gbps,
                 100 * v.gbps / med, med);
      } else if (med > 0 && v.gbps < rule.warn_ratio * med) {
        v.health = 10;
        snprintf(v.message, sizeof v.message, "%.1f GB/s is %.0f %% of the other links' median %.1f GB/s", v.gbps,
                 100 * v.gbps / med, med);
      } else {
        snprintf(v.message, sizeof v.message, "%.1f GB/s", v.gbps);
      }
      worst = std::max(worst, (int)v.health);
      out[(*n_out)++] = v;
    }
  return worst;
}

}  // extern "C"
