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
#include <cstdarg>
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

// The copy of one (initiator, peer) pair.  `from` and `to` are plain global pointers: with a local source and a
// destination in the peer's memory the lanes store over the link (push), with a source in the peer's memory and a
// local destination they load over it (pull).  Each workgroup walks one contiguous span so that a wave's accesses
// are 1 KiB runs and the link sees long bursts; four independent 16-byte loads are in flight per lane before the
// first store.
__device__ __forceinline__ void copy_span(const u32x4* __restrict__ from, u32x4* __restrict__ to, uint64_t begin, uint64_t end) {
  uint64_t w = begin + threadIdx.x;
  for (; w + 3 * 256 < end; w += 4 * 256) {
    u32x4 a = __builtin_nontemporal_load(from + w);
    u32x4 b = __builtin_nontemporal_load(from + w + 256);
    u32x4 c = __builtin_nontemporal_load(from + w + 512);
    u32x4 d = __builtin_nontemporal_load(from + w + 768);
    __builtin_nontemporal_store(a, to + w);
    __builtin_nontemporal_store(b, to + w + 256);
    __builtin_nontemporal_store(c, to + w + 512);
    __builtin_nontemporal_store(d, to + w + 768);
  }
  for (; w < end; w += 256) __builtin_nontemporal_store(__builtin_nontemporal_load(from + w), to + w);
}

// One launch drives every peer of a GPU at once.  A stream per peer cannot: a process has four hardware queues
// on a GPU, so of seven per-peer streams three would wait behind the others and each pair would be timed under a
// different load.  Here workgroup b serves peer b % npeers (neighbouring workgroups start on different links)
// and is the (b / npeers)-th of that peer's `wgs` workgroups.  Per peer the launch leaves two stamps of the
// constant-rate real-time counter, which all compute units of a GPU share: the earliest entry and the latest exit
// (after the workgroup's stores were acknowledged) of that peer's workgroups.  bytes * iters over (stop - start)
// is then the rate of that one link while all the others were busy.
constexpr int kMaxFan = GMD_LINK_MAX_PEERS;
struct FanPeer {
  const u32x4* from;
  u32x4* to;
};
struct FanArgs {
  FanPeer peer[kMaxFan];
  uint64_t words;
  int npeers, wgs, iters;
};
struct FanStamps {
  unsigned long long start[kMaxFan];  // preset to ~0
  unsigned long long stop[kMaxFan];   // preset to 0
};

__global__ __launch_bounds__(256) void link_fan_kernel(FanArgs a, FanStamps* stamps) {
  const int p = (int)(blockIdx.x % (unsigned)a.npeers), g = (int)(blockIdx.x / (unsigned)a.npeers);
  if (threadIdx.x == 0) atomicMin(&stamps->start[p], (unsigned long long)__builtin_amdgcn_s_memrealtime());
  const uint64_t per = (a.words + a.wgs - 1) / a.wgs;
  const uint64_t begin = (uint64_t)g * per;
  const uint64_t end = begin + per < a.words ? begin + per : a.words;
  const u32x4* from = a.peer[p].from;
  u32x4* to = a.peer[p].to;
  for (int it = 0; it < a.iters; it++)
    if (begin < end) copy_span(from, to, begin, end);
  __threadfence_system();  // the stores have left this GPU's caches and were acknowledged
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(&stamps->stop[p], (unsigned long long)__builtin_amdgcn_s_memrealtime());
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

// Ticks per second of s_memrealtime on this device (100 MHz on MI355X; asked, not assumed).
int realtime_hz(int device, double* hz) {
  int khz = 0;
  GMD_HIP(hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, device));
  *hz = khz > 0 ? khz * 1e3 : 100e6;
  return GMD_OK;
}

// One fused launch on one GPU: a stream, an event pair around the timed launch, the stamps.
struct Fan {
  int device = -1;
  hipStream_t stream = nullptr;
  hipEvent_t start = nullptr, stop = nullptr;
  FanStamps* stamps = nullptr;  // device memory
  double hz = 100e6;
  FanArgs args = {};
  std::vector<int> peer_pos;  // which device-list position each entry of args.peer is
  bool queued = false;

  ~Fan() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    if (start) (void)hipEventDestroy(start);
    if (stop) (void)hipEventDestroy(stop);
    if (stream) (void)hipStreamDestroy(stream);
    if (stamps) (void)hipFree(stamps);
  }
  int setup(int dev) {
    device = dev;
    GMD_HIP(hipSetDevice(dev));
    GMD_HIP(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    GMD_HIP(hipEventCreate(&start));
    GMD_HIP(hipEventCreate(&stop));
    GMD_HIP(hipMalloc((void**)&stamps, sizeof(FanStamps)));
    return realtime_hz(dev, &hz);
  }
  void clear() {
    memset(&args, 0, sizeof args);
    peer_pos.clear();
    queued = false;
  }
  bool add(int pos, const u32x4* from, u32x4* to) {
    if (args.npeers >= kMaxFan) return false;
    args.peer[args.npeers].from = from;
    args.peer[args.npeers].to = to;
    args.npeers++;
    peer_pos.push_back(pos);
    return true;
  }
  // A one-pass warm-up launch (page tables of the peer mappings, instruction cache), then the timed launch of
  // `iters` passes between the two events.  Asynchronous.
  int enqueue(uint64_t words, int wgs, int iters) {
    if (!args.npeers) return GMD_OK;
    GMD_HIP(hipSetDevice(device));
    args.words = words, args.wgs = wgs, args.iters = 1;
    const unsigned grid = (unsigned)(args.npeers * wgs);
    link_fan_kernel<<<grid, 256, 0, stream>>>(args, stamps);
    GMD_HIP(hipMemsetAsync(stamps->start, 0xff, sizeof stamps->start, stream));
    GMD_HIP(hipMemsetAsync(stamps->stop, 0, sizeof stamps->stop, stream));
    args.iters = iters;
    GMD_HIP(hipEventRecord(start, stream));
    link_fan_kernel<<<grid, 256, 0, stream>>>(args, stamps);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipEventRecord(stop, stream));
    queued = true;
    return GMD_OK;
  }
  // Waits for the launch.  gbps[k] = rate of args.peer[k] inside its own stamp window; *launch_ms = the event
  // pair's time; window_ms (may be null) = per-peer window lengths.
  int collect(std::vector<double>* gbps, double* launch_ms, std::vector<double>* window_ms) {
    gbps->assign(args.npeers, 0);
    if (window_ms) window_ms->assign(args.npeers, 0);
    *launch_ms = 0;
    if (!queued) return GMD_OK;
    GMD_HIP(hipSetDevice(device));
    GMD_HIP(hipEventSynchronize(stop));
    float ms = 0;
    GMD_HIP(hipEventElapsedTime(&ms, start, stop));
    *launch_ms = ms;
    FanStamps h;
    GMD_HIP(hipMemcpy(&h, stamps, sizeof h, hipMemcpyDeviceToHost));
    for (int k = 0; k < args.npeers; k++) {
      if (h.stop[k] <= h.start[k]) continue;  // stamps not written: leaves 0, which grades as a failure
      const double s = (double)(h.stop[k] - h.start[k]) / hz;
      (*gbps)[k] = (double)args.words * 16 * args.iters / s / 1e9;
      if (window_ms) (*window_ms)[k] = s * 1e3;
    }
    queued = false;
    return GMD_OK;
  }
};

struct LinkRig {
  std::vector<int> ids;
  uint64_t words = 0;
  int iters = 5, wgs = 64;
  bool pull = false;
  std::vector<void*> src;               // per GPU: the buffer it sends (push) or lets its peers read (pull)
  std::vector<void*> dst;               // per GPU: ndev slots of `words`; push: slot s receives from position s,
                                        // pull: slot s is where this GPU puts what it read from position s
  std::vector<unsigned long long*> bad; // per GPU: one counter
  std::vector<char> access;             // [i*n+j]: i can address j's memory
  std::vector<Fan> fans;                // per GPU

  ~LinkRig() {
    fans.clear();
    const int n = (int)ids.size();
    for (int i = 0; i < n; i++) {
      (void)hipSetDevice(ids[i]);
      if (i < (int)src.size() && src[i]) (void)hipFree(src[i]);
      if (i < (int)dst.size() && dst[i]) (void)hipFree(dst[i]);
      if (i < (int)bad.size() && bad[i]) (void)hipFree(bad[i]);
    }
  }

  u32x4* slot(int owner, int s) const { return (u32x4*)dst[owner] + (uint64_t)s * words; }
  bool ok(int i, int j) const { return i != j && access[(size_t)i * ids.size() + j]; }

  int setup() {
    const int n = (int)ids.size();
    src.assign(n, nullptr), dst.assign(n, nullptr), bad.assign(n, nullptr);
    access.assign((size_t)n * n, 0);
    fans.resize(n);
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
        if (ids[i] == ids[j]) {  // one card listed twice (rehearsal on a single GPU): its own memory stands in for the peer's
          access[(size_t)i * n + j] = 1;
          continue;
        }
        int can = 0;
        GMD_HIP(hipDeviceCanAccessPeer(&can, ids[i], ids[j]));
        if (!can) continue;
        hipError_t e = hipDeviceEnablePeerAccess(ids[j], 0);
        (void)hipGetLastError();
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) continue;
        access[(size_t)i * n + j] = 1;
      }
      int rc = fans[i].setup(ids[i]);
      if (rc) return rc;
      GMD_HIP(hipDeviceSynchronize());
    }
    return GMD_OK;
  }

  // Adds pair (initiator i, peer j) to i's next launch.
  void add(int i, int j) {
    if (pull) fans[i].add(j, (const u32x4*)src[j], slot(i, j));
    else fans[i].add(j, (const u32x4*)src[i], slot(j, i));
  }

  // Launches what was added on every GPU (back to back: the GPUs run concurrently), waits, writes the cells.
  int run(double* gbps, double* per_gpu_gbps) {
    const int n = (int)ids.size();
    int rc;
    for (int i = 0; i < n; i++)
      if ((rc = fans[i].enqueue(words, wgs, iters))) return rc;
    for (int i = 0; i < n; i++) {
      std::vector<double> g;
      double ms = 0;
      const int npeers = fans[i].args.npeers;
      if ((rc = fans[i].collect(&g, &ms, nullptr))) return rc;
      for (int k = 0; k < npeers; k++) gbps[(size_t)i * n + fans[i].peer_pos[k]] = g[k];
      if (per_gpu_gbps && npeers) per_gpu_gbps[i] = ms > 0 ? (double)words * 16 * iters * npeers / (ms * 1e-3) / 1e9 : 0;
    }
    return GMD_OK;
  }

  // Data check of pair (initiator i, peer j) on the GPU that holds the destination, after the launches drained.
  int check(int i, int j, uint64_t* out) {
    const int owner = pull ? i : j, from = pull ? j : i, s = pull ? j : i;
    GMD_HIP(hipSetDevice(ids[owner]));
    GMD_HIP(hipMemset(bad[owner], 0, sizeof(unsigned long long)));
    link_check_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>(slot(owner, s), words, (uint32_t)from, bad[owner]);
    GMD_HIP(hipGetLastError());
    unsigned long long v = 0;
    GMD_HIP(hipMemcpy(&v, bad[owner], sizeof v, hipMemcpyDeviceToHost));
    *out = v;
    // the next pattern must not pass on this pattern's data
    GMD_HIP(hipMemset(slot(owner, s), 0, words * 16));
    return GMD_OK;
  }
};

std::string sf(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
std::string sf(const char* fmt, ...) {
  char b[256];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(b, sizeof b, fmt, ap);
  va_end(ap);
  return b;
}

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
  if (n <= 0 || n > GMD_LINK_MAX_PEERS + 1) {
    gmd::set_error("%d GPUs requested (1..%d)", n, GMD_LINK_MAX_PEERS + 1);
    return GMD_ERR_BADPARAM;
  }
  LinkRig rig;
  // GPUMON_DIAG_ALLOW_REPEAT=1 lets a card stand in several positions, so that every pattern, the per-position
  // bookkeeping and the grading can be run through on a machine with one GPU (its links then are its own memory).
  const bool repeat_ok = getenv("GPUMON_DIAG_ALLOW_REPEAT") != nullptr;
  if (n > visible && !(repeat_ok && cfg.devices)) {
    gmd::set_error("%d GPUs requested, %d visible", n, visible);
    return GMD_ERR_BADPARAM;
  }
  for (int i = 0; i < n; i++) {
    const int id = cfg.devices ? cfg.devices[i] : i;
    if (id < 0 || id >= visible || (!repeat_ok && std::find(rig.ids.begin(), rig.ids.end(), id) != rig.ids.end())) {
      gmd::set_error("bad or repeated device index %d", id);
      return GMD_ERR_BADPARAM;
    }
    rig.ids.push_back(id);
  }
  for (int i = 0; i < n * n; i++) {
    gbps[i] = 0;
    if (bad) bad[i] = 0;
  }
  for (int i = 0; i < n && cfg.per_gpu_gbps; i++) cfg.per_gpu_gbps[i] = 0;
  if (n == 1) return GMD_OK;  // no link to test

  const uint64_t bytes = cfg.bytes ? cfg.bytes : 256ull << 20;
  rig.words = std::max<uint64_t>(256, bytes / 16);
  rig.iters = cfg.iters > 0 ? cfg.iters : 5;
  rig.wgs = cfg.workgroups > 0 ? std::min(cfg.workgroups, 1024) : 64;
  rig.pull = cfg.pull != 0;
  int rc = rig.setup();
  if (rc) return rc;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (i != j && !rig.ok(i, j)) gbps[i * n + j] = -1;

  auto verify = [&](int i, int j) -> int {
    uint64_t b = 0;
    int r = cfg.verify ? rig.check(i, j, &b) : GMD_OK;
    if (bad) bad[i * n + j] = b;
    return r;
  };
  auto clear = [&] {
    for (auto& f : rig.fans) f.clear();
  };

  if (mode == GMD_LINK_UNI) {
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        if (!rig.ok(i, j)) continue;
        clear();
        rig.add(i, j);
        if ((rc = rig.run(gbps, nullptr)) || (rc = verify(i, j))) return rc;
      }
  } else if (mode == GMD_LINK_BIDI) {
    for (int i = 0; i < n; i++)
      for (int j = i + 1; j < n; j++) {
        if (!rig.ok(i, j) || !rig.ok(j, i)) continue;
        clear();
        rig.add(i, j);
        rig.add(j, i);
        if ((rc = rig.run(gbps, nullptr)) || (rc = verify(i, j)) || (rc = verify(j, i))) return rc;
      }
  } else {
    // One launch per GPU that serves all of its peers: seven links of a GPU busy at once out of one hardware queue.
    clear();
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (rig.ok(i, j)) rig.add(i, j);
    if ((rc = rig.run(gbps, cfg.per_gpu_gbps))) return rc;
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++)
        if (rig.ok(i, j) && (rc = verify(i, j))) return rc;
  }
  return GMD_OK;
}

int gmd_link_selftest(int device, uint64_t bytes, int workgroups, double* gbps, uint64_t* bad) {
  if (!gbps || !bad) return GMD_ERR_BADPARAM;
  gmd_link_fan_result_t r;
  return gmd_link_fan_selftest(device, 1, bytes, workgroups, 5, gbps, bad, &r);
}

int gmd_link_fan_selftest(int device, int npeers, uint64_t bytes, int workgroups, int iters, double* gbps, uint64_t* bad,
                          gmd_link_fan_result_t* res) {
  if (!gbps || !bad || !res || npeers < 1 || npeers > GMD_LINK_MAX_PEERS) return GMD_ERR_BADPARAM;
  memset(res, 0, sizeof *res);
  const uint64_t words = std::max<uint64_t>(1, (bytes ? bytes : 256ull << 20) / 16);
  const int wgs = workgroups > 0 ? std::min(workgroups, 1024) : 64;
  if (iters <= 0) iters = 5;
  GMD_HIP(hipSetDevice(device));
  // One source per stand-in peer (its own pattern) and one destination each, with one guard word past the end
  // that the kernel must leave alone.
  std::vector<gmd::DeviceBuffer> src(npeers), dst(npeers);
  gmd::DeviceBuffer cnt;
  int rc;
  if ((rc = cnt.alloc(sizeof(unsigned long long)))) return rc;
  Fan fan;
  if ((rc = fan.setup(device))) return rc;
  fan.clear();
  for (int k = 0; k < npeers; k++) {
    if ((rc = src[k].alloc(words * 16)) || (rc = dst[k].alloc((words + 1) * 16))) return rc;
    link_fill_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>((u32x4*)src[k].p, words, 5u + k);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipMemset(dst[k].p, 0xa5, (words + 1) * 16));
    fan.add(k, (const u32x4*)src[k].p, (u32x4*)dst[k].p);
  }
  GMD_HIP(hipDeviceSynchronize());
  if ((rc = fan.enqueue(words, wgs, iters))) return rc;
  std::vector<double> g, win;
  double ms = 0;
  if ((rc = fan.collect(&g, &ms, &win))) return rc;
  res->npeers = npeers;
  res->launch_ms = ms;
  res->tick_hz = fan.hz;
  res->window_ms_min = 1e300;
  for (int k = 0; k < npeers; k++) {
    gbps[k] = g[k];
    res->window_ms_min = std::min(res->window_ms_min, win[k]);
    res->window_ms_max = std::max(res->window_ms_max, win[k]);
  }
  res->launch_gbps = ms > 0 ? (double)words * 16 * iters * npeers / (ms * 1e-3) / 1e9 : 0;
  for (int k = 0; k < npeers; k++) {
    GMD_HIP(hipMemset(cnt.p, 0, sizeof(unsigned long long)));
    link_check_kernel<<<gmd::stream_grid(words, 256, 4), 256>>>((const u32x4*)dst[k].p, words, 5u + k, (unsigned long long*)cnt.p);
    GMD_HIP(hipGetLastError());
    unsigned long long v = 0;
    uint32_t guard[4];
    GMD_HIP(hipMemcpy(&v, cnt.p, sizeof v, hipMemcpyDeviceToHost));
    GMD_HIP(hipMemcpy(guard, (char*)dst[k].p + words * 16, 16, hipMemcpyDeviceToHost));
    for (uint32_t x : guard)
      if (x != 0xa5a5a5a5u) v++;
    bad[k] = v;
  }
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

void gmd_link_expect_defaults(gmd_link_expect_t* e) {
  if (!e) return;
  if (e->link_gbps <= 0) e->link_gbps = 76.8;  // MI355X: 153.6 GB/s per link, both directions together
  if (e->derate_uni <= 0) e->derate_uni = 0.70;
  if (e->derate_bidi <= 0) e->derate_bidi = 0.65;
  if (e->derate_all <= 0) e->derate_all = 0.60;
  if (e->pull_factor <= 0) e->pull_factor = 0.80;
  if (e->derate_collective <= 0) e->derate_collective = 0.50;
  if (e->warn_ratio <= 0) e->warn_ratio = 0.90;
  if (e->fail_ratio <= 0) e->fail_ratio = 0.75;
  if (e->fail_ratio > e->warn_ratio) e->fail_ratio = e->warn_ratio;
  if (e->scale <= 0) e->scale = 1.0;
}

double gmd_link_expected_gbps(const gmd_link_expect_t* e_in, int mode, int peers) {
  gmd_link_expect_t e;
  memset(&e, 0, sizeof e);
  if (e_in) e = *e_in;
  gmd_link_expect_defaults(&e);
  const double d = mode == GMD_LINK_UNI ? e.derate_uni : mode == GMD_LINK_BIDI ? e.derate_bidi : e.derate_all;
  return e.link_gbps * d * (e.pull ? e.pull_factor : 1.0) * e.scale * (peers > 0 ? peers : 1);
}

double gmd_rccl_expected_busbw(const gmd_link_expect_t* e_in, int op, int ndev) {
  if (ndev < 2 || op < 0 || op >= GMD_COLL_COUNT) return 0;
  gmd_link_expect_t e;
  memset(&e, 0, sizeof e);
  if (e_in) e = *e_in;
  gmd_link_expect_defaults(&e);
  switch (op) {
    // A ring moves every byte over one link per hop: whatever the library builds on top of the mesh, it must not
    // be slower than one ring on one link.
    case GMD_COLL_ALL_REDUCE:
    case GMD_COLL_BROADCAST: return e.link_gbps * e.derate_uni * e.scale;
    // Every GPU has something for every peer: all n-1 links of a GPU can carry data at once.
    default: return e.link_gbps * (ndev - 1) * e.derate_collective * e.scale;
  }
}

int gmd_grade_against(double value, double expected, double warn_ratio, double fail_ratio) {
  if (!(expected > 0)) return 0;
  if (!std::isfinite(value) || value < fail_ratio * expected) return 20;
  return value < warn_ratio * expected ? 10 : 0;
}

int gmd_outlier_grade(int n, const double* values, double warn_ratio, double fail_ratio, int32_t* health, double* others_median) {
  if (n < 1 || !values || !health) return GMD_ERR_BADPARAM;
  if (warn_ratio <= 0) warn_ratio = 0.95;
  if (fail_ratio <= 0) fail_ratio = 0.90;
  int worst = 0;
  for (int i = 0; i < n; i++) {
    std::vector<double> others;
    for (int k = 0; k < n; k++)
      if (k != i && std::isfinite(values[k]) && values[k] > 0) others.push_back(values[k]);
    const double med = others.empty() ? 0 : median_of(others);
    if (others_median) others_median[i] = med;
    health[i] = others.empty() ? 0 : gmd_grade_against(values[i], med, warn_ratio, fail_ratio);
    worst = std::max(worst, (int)health[i]);
  }
  return worst;
}

const char* gmd_link_finding_name(int kind) {
  static const char* names[] = {"no_access", "corrupt_data", "slow_link", "half_duplex", "slow_under_load", "low_egress", "low_ingress"};
  return kind >= 0 && kind < GMD_LINK_FINDING_KINDS ? names[kind] : "unknown";
}

int gmd_link_diagnose(int n, const double* uni, const double* bidi, const double* all, const uint64_t* bad,
                      const gmd_link_expect_t* e_in, gmd_link_finding_t* out, int* n_out) {
  if (n < 1 || !uni || !out || !n_out) return GMD_ERR_BADPARAM;
  gmd_link_expect_t e;
  memset(&e, 0, sizeof e);
  if (e_in) e = *e_in;
  gmd_link_expect_defaults(&e);
  const int cap = *n_out;
  *n_out = 0;
  int worst = 0;
  bool overflow = false;
  auto emit = [&](int kind, int mode, int src, int dst, int health, double got, double want, const std::string& msg) {
    if (*n_out >= cap) {
      overflow = true;
      return;
    }
    gmd_link_finding_t f;
    memset(&f, 0, sizeof f);
    f.kind = kind, f.mode = mode, f.src = src, f.dst = dst, f.health = health, f.gbps = got, f.expected_gbps = want;
    snprintf(f.message, sizeof f.message, "%s", msg.c_str());
    worst = std::max(worst, health);
    out[(*n_out)++] = f;
  };
  auto at = [&](int i, int j) { return (size_t)i * n + j; };
  auto measured = [&](const double* m, int i, int j) { return i != j && std::isfinite(m[at(i, j)]) && m[at(i, j)] >= 0; };
  // 1 = the pair has a finding from the one-at-a-time pattern (left out of everything that follows), 2 = from a busier one
  std::vector<char> flagged((size_t)n * n, 0);

  // 1. one pair at a time: reachability, data, the rate of one direction of one link
  std::vector<double> pool;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++)
      if (measured(uni, i, j)) pool.push_back(uni[at(i, j)]);
  const double want_uni = gmd_link_expected_gbps(&e, GMD_LINK_UNI, 1);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) {
      if (i == j) continue;
      const double g = uni[at(i, j)];
      if (!measured(uni, i, j)) {
        flagged[at(i, j)] = 1;
        emit(GMD_LINK_NO_ACCESS, GMD_LINK_UNI, i, j, 20, g, want_uni, "no peer access");
        continue;
      }
      if (bad && bad[at(i, j)]) {
        flagged[at(i, j)] = 1;
        emit(GMD_LINK_CORRUPT, GMD_LINK_UNI, i, j, 20, g, want_uni, sf("%llu corrupt 16-byte words", (unsigned long long)bad[at(i, j)]));
        continue;
      }
      std::vector<double> others = pool;
      others.erase(std::find(others.begin(), others.end(), g));
      const double med = others.size() >= 2 ? median_of(others) : 0;
      const int h_abs = gmd_grade_against(g, want_uni, e.warn_ratio, e.fail_ratio);
      const int h_rel = gmd_grade_against(g, med, e.warn_ratio, e.fail_ratio);
      if (!h_abs && !h_rel) continue;
      flagged[at(i, j)] = 1;
      if (h_rel >= h_abs)
        emit(GMD_LINK_SLOW, GMD_LINK_UNI, i, j, h_rel, g, med, sf("%.1f GB/s is %.0f %% of the other links' median %.1f GB/s", g, 100 * g / med, med));
      else
        emit(GMD_LINK_SLOW, GMD_LINK_UNI, i, j, h_abs, g, want_uni,
             sf("%.1f GB/s is %.0f %% of the %.1f GB/s expected of one direction of a %.1f GB/s link", g, 100 * g / want_uni, want_uni, e.link_gbps));
    }

  // 2. both directions at once: a direction that loses more than the tolerance against what it carries alone
  if (bidi) {
    const double want_bidi = gmd_link_expected_gbps(&e, GMD_LINK_BIDI, 1);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        if (i == j || flagged[at(i, j)] == 1 || flagged[at(j, i)] == 1 || !measured(bidi, i, j)) continue;
        const double g = bidi[at(i, j)], alone = uni[at(i, j)];
        const int h_rel = gmd_grade_against(g, alone, e.warn_ratio, e.fail_ratio);
        const int h_abs = gmd_grade_against(g, want_bidi, e.warn_ratio, e.fail_ratio);
        if (!h_rel && !h_abs) continue;
        flagged[at(i, j)] = 2;
        if (h_rel >= h_abs)
          emit(GMD_LINK_HALF_DUPLEX, GMD_LINK_BIDI, i, j, h_rel, g, alone,
               sf("%.1f GB/s with both directions busy, %.0f %% of the %.1f GB/s it carries alone", g, 100 * g / alone, alone));
        else
          emit(GMD_LINK_HALF_DUPLEX, GMD_LINK_BIDI, i, j, h_abs, g, want_bidi,
               sf("%.1f GB/s with both directions busy, %.0f %% of the %.1f GB/s expected", g, 100 * g / want_bidi, want_bidi));
      }
  }

  // 3. every GPU on every peer at once.  First what each GPU moves over its healthy links as initiator (row) and as
  //    peer (column): the worst outlier is reported and its cells set aside, then the rest is looked at again, so
  //    that one bad port does not drag its seven neighbours under the tolerance with it.  Then single pairs that
  //    fall behind their GPU's other links only under load.
  if (all && n > 2) {
    const double want_all = gmd_link_expected_gbps(&e, GMD_LINK_ALL, 1);
    std::vector<char> skip((size_t)n * n, 0);
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) skip[at(i, j)] = i == j || flagged[at(i, j)] || !measured(all, i, j);
    std::vector<char> row_done(n, 0), col_done(n, 0);
    for (int round = 0; round < 2 * n; round++) {
      std::vector<double> mean[2] = {std::vector<double>(n, 0), std::vector<double>(n, 0)};
      std::vector<int> cnt[2] = {std::vector<int>(n, 0), std::vector<int>(n, 0)};
      for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
          if (skip[at(i, j)]) continue;
          mean[0][i] += all[at(i, j)], cnt[0][i]++;
          mean[1][j] += all[at(i, j)], cnt[1][j]++;
        }
      int pick_side = -1, pick = -1, pick_h = 0;
      double pick_ratio = 1e300, pick_ref = 0;
      bool pick_rel = false;
      for (int side = 0; side < 2; side++) {
        for (int g = 0; g < n; g++) mean[side][g] = cnt[side][g] ? mean[side][g] / cnt[side][g] : NAN;
        for (int g = 0; g < n; g++) {
          if (!cnt[side][g] || (side ? col_done[g] : row_done[g])) continue;
          std::vector<double> others;
          for (int k = 0; k < n; k++)
            if (k != g && cnt[side][k]) others.push_back(mean[side][k]);
          const double med = others.empty() ? 0 : median_of(others);
          const int h_rel = gmd_grade_against(mean[side][g], med, e.warn_ratio, e.fail_ratio);
          const int h_abs = gmd_grade_against(mean[side][g], want_all, e.warn_ratio, e.fail_ratio);
          if (!h_rel && !h_abs) continue;
          const bool rel = h_rel >= h_abs;
          const double ref = rel ? med : want_all, ratio = mean[side][g] / ref;
          if (ratio < pick_ratio) pick_ratio = ratio, pick_side = side, pick = g, pick_h = std::max(h_rel, h_abs), pick_ref = ref, pick_rel = rel;
        }
      }
      if (pick < 0) break;
      const int links = cnt[pick_side][pick];
      const double total = mean[pick_side][pick] * links;
      const char* verb = pick_side == 0 ? (e.pull ? "reads from" : "writes to") : (e.pull ? "is read by" : "is written by");
      const std::string msg =
          pick_rel ? sf("%s %d peers at once at %.1f GB/s in total, %.0f %% of the other GPUs' median %.1f GB/s", verb, links, total, 100 * pick_ratio, pick_ref * links)
                   : sf("%s %d peers at once at %.1f GB/s in total, %.0f %% of the %.1f GB/s expected of %d links", verb, links, total, 100 * pick_ratio, pick_ref * links, links);
      if (pick_side == 0) emit(GMD_LINK_LOW_EGRESS, GMD_LINK_ALL, pick, -1, pick_h, total, pick_ref * links, msg);
      else emit(GMD_LINK_LOW_INGRESS, GMD_LINK_ALL, -1, pick, pick_h, total, pick_ref * links, msg);
      (pick_side ? col_done : row_done)[pick] = 1;
      for (int k = 0; k < n; k++) skip[pick_side == 0 ? at(pick, k) : at(k, pick)] = 1;
    }
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        if (skip[at(i, j)] || flagged[at(i, j)]) continue;
        std::vector<double> sib;
        for (int k = 0; k < n; k++)
          if (k != j && !skip[at(i, k)]) sib.push_back(all[at(i, k)]);
        if (sib.size() < 2) continue;
        const double med = median_of(sib), g = all[at(i, j)];
        const int h = gmd_grade_against(g, med, e.warn_ratio, e.fail_ratio);
        if (h)
          emit(GMD_LINK_SLOW_UNDER_LOAD, GMD_LINK_ALL, i, j, h, g, med,
               sf("%.1f GB/s with every link busy, %.0f %% of the median %.1f GB/s of this GPU's other links", g, 100 * g / med, med));
      }
  }
  if (overflow) return GMD_ERR_BADPARAM;
  return worst;
}

}  // extern "C"
