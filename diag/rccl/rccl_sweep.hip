// Interconnect diagnostics for one MI355X node, driven from a single process:
// an RCCL collective sweep (all-reduce, all-gather, reduce-scatter, all-to-all,
// broadcast; 8 B .. 1 GiB) and a pairwise peer-copy matrix.
//
// The 8 GPUs of an MI355X node form a full mesh of xGMI links.  A ring
// collective moves data over one link per hop, so its bus bandwidth is bound by a
// single link, while all-to-all and the pairwise matrix use every link; reporting
// both separates "one slow link" from "slow fabric".  Every size is first run on
// patterned data and checked on the device, so a link that corrupts data fails
// the sweep instead of merely slowing it.
//
// librccl.so is resolved with dlopen at first use: the diagnostics library stays
// loadable on hosts that have the HIP runtime but no RCCL.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <chrono>
#include <mutex>

#include "../hip/diag_common.h"

namespace {

#define RCCL_FUNCS(X)     \
  X(ncclGetErrorString)   \
  X(ncclCommInitAll)      \
  X(ncclCommDestroy)      \
  X(ncclCommCount)        \
  X(ncclCommUserRank)     \
  X(ncclCommCuDevice)     \
  X(ncclGroupStart)       \
  X(ncclGroupEnd)         \
  X(ncclAllReduce)        \
  X(ncclAllGather)        \
  X(ncclReduceScatter)    \
  X(ncclAllToAll)         \
  X(ncclBroadcast)

struct Rccl {
  void* handle = nullptr;
#define X(name) decltype(&::name) name = nullptr;
  RCCL_FUNCS(X)
#undef X
};

Rccl* load_rccl() {
  static Rccl r;
  static std::once_flag once;
  static bool ok = false;
  std::call_once(once, [] {
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names)
      if ((r.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!r.handle) {
      gmd::set_error("librccl.so not found: %s", dlerror());
      return;
    }
    ok = true;
#define X(name)                                                  \
  r.name = (decltype(r.name))dlsym(r.handle, #name);             \
  if (!r.name) {                                                 \
    gmd::set_error("librccl.so lacks %s", #name);                \
    ok = false;                                                  \
  }
    RCCL_FUNCS(X)
#undef X
  });
  return ok ? &r : nullptr;
}

#define GMD_NCCL(R, expr)                                                                   \
  do {                                                                                      \
    ncclResult_t _r = (expr);                                                               \
    if (_r != ncclSuccess) {                                                                \
      gmd::set_error("%s failed: %s (%s:%d)", #expr, (R)->ncclGetErrorString(_r), __FILE__, __LINE__); \
      return GMD_ERR_HIP;                                                                   \
    }                                                                                       \
  } while (0)

// Patterned data: element i of rank r's logical buffer is ((i + off) % 7) + r —
// small integers, so sums over up to 16 ranks are exact in fp32.
__global__ void fill_kernel(float* p, uint64_t n, uint64_t off, int rank) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = (float)((i + off) % 7) + (float)rank;
}

__global__ void check_kernel(const float* p, uint64_t n, int op, int rank, int world, uint64_t shard,
                             unsigned long long* bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long mine = 0;
  const float tri = (float)(world * (world - 1) / 2);
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float want;
    switch (op) {
      case GMD_COLL_ALL_REDUCE: want = (float)(i % 7) * world + tri; break;
      case GMD_COLL_ALL_GATHER: want = (float)(i % 7) + (float)(i / shard); break;
      case GMD_COLL_REDUCE_SCATTER: want = (float)((i + rank * shard) % 7) * world + tri; break;
      case GMD_COLL_ALL_TO_ALL: want = (float)((i % shard + rank * shard) % 7) + (float)(i / shard); break;
      default: want = (float)(i % 7); break;  // broadcast from rank 0
    }
    if (p[i] != want) mine++;
  }
  if (mine) atomicAdd(bad, mine);
}

struct Dev {
  int id = 0;
  hipStream_t stream = nullptr;
  hipEvent_t t0 = nullptr, t1 = nullptr;
  float* send = nullptr;
  float* recv = nullptr;
  unsigned long long* bad = nullptr;
};

struct Sweep {
  Rccl* R;
  std::vector<Dev> devs;
  std::vector<ncclComm_t> comms;
  ~Sweep() {
    for (auto& c : comms)
      if (c) R->ncclCommDestroy(c);
    for (auto& d : devs) {
      (void)hipSetDevice(d.id);
      if (d.stream) (void)hipStreamDestroy(d.stream);
      if (d.t0) (void)hipEventDestroy(d.t0);
      if (d.t1) (void)hipEventDestroy(d.t1);
      if (d.send) (void)hipFree(d.send);
      if (d.recv) (void)hipFree(d.recv);
      if (d.bad) (void)hipFree(d.bad);
    }
  }

  // Device-side clock of a timed run: an event on every GPU's stream before and after, the slowest GPU counts.
  // (A host clock around the calls would add the cost of enqueueing to every row.  Rows of less than about
  // 1 MiB are still bound by launch latency, not by the links: `iters` grouped launches take what the host
  // needs to issue them.)
  int mark(bool begin) {
    for (auto& d : devs) {
      GMD_HIP(hipSetDevice(d.id));
      GMD_HIP(hipEventRecord(begin ? d.t0 : d.t1, d.stream));
    }
    return GMD_OK;
  }
  int elapsed_s(double* s) {
    *s = 0;
    for (auto& d : devs) {
      GMD_HIP(hipSetDevice(d.id));
      GMD_HIP(hipEventSynchronize(d.t1));
      float ms = 0;
      GMD_HIP(hipEventElapsedTime(&ms, d.t0, d.t1));
      *s = std::max(*s, (double)ms * 1e-3);
    }
    return GMD_OK;
  }

  int sync_all() {
    for (auto& d : devs) {
      GMD_HIP(hipSetDevice(d.id));
      GMD_HIP(hipStreamSynchronize(d.stream));
    }
    return GMD_OK;
  }

  // n = element count of the payload (multiple of world for the sharded ops).
  int launch(int op, uint64_t n) {
    const int w = (int)devs.size();
    const uint64_t shard = n / w;
    GMD_NCCL(R, R->ncclGroupStart());
    for (int r = 0; r < w; r++) {
      Dev& d = devs[r];
      switch (op) {
        case GMD_COLL_ALL_REDUCE: GMD_NCCL(R, R->ncclAllReduce(d.send, d.recv, n, ncclFloat, ncclSum, comms[r], d.stream)); break;
        case GMD_COLL_ALL_GATHER: GMD_NCCL(R, R->ncclAllGather(d.send, d.recv, shard, ncclFloat, comms[r], d.stream)); break;
        case GMD_COLL_REDUCE_SCATTER: GMD_NCCL(R, R->ncclReduceScatter(d.send, d.recv, shard, ncclFloat, ncclSum, comms[r], d.stream)); break;
        case GMD_COLL_ALL_TO_ALL: GMD_NCCL(R, R->ncclAllToAll(d.send, d.recv, shard, ncclFloat, comms[r], d.stream)); break;
        default: GMD_NCCL(R, R->ncclBroadcast(d.send, d.recv, n, ncclFloat, 0, comms[r], d.stream)); break;
      }
    }
    GMD_NCCL(R, R->ncclGroupEnd());
    return GMD_OK;
  }

  int verify(int op, uint64_t n, bool* ok) {
    const int w = (int)devs.size();
    const uint64_t shard = n / w;
    for (int r = 0; r < w; r++) {
      Dev& d = devs[r];
      GMD_HIP(hipSetDevice(d.id));
      const bool sharded_in = op == GMD_COLL_ALL_GATHER;
      const uint64_t n_in = sharded_in ? shard : n;
      const unsigned grid = gmd::stream_grid(n_in, 256, 4);
      fill_kernel<<<grid, 256, 0, d.stream>>>(d.send, n_in, sharded_in ? r * shard : 0, r);
      GMD_HIP(hipGetLastError());
      GMD_HIP(hipMemsetAsync(d.bad, 0, sizeof(unsigned long long), d.stream));
    }
    int rc = launch(op, n);
    if (rc) return rc;
    *ok = true;
    for (int r = 0; r < w; r++) {
      Dev& d = devs[r];
      GMD_HIP(hipSetDevice(d.id));
      const uint64_t n_out = op == GMD_COLL_REDUCE_SCATTER ? shard : n;
      check_kernel<<<gmd::stream_grid(n_out, 256, 4), 256, 0, d.stream>>>(d.recv, n_out, op, r, w, shard, d.bad);
      GMD_HIP(hipGetLastError());
      unsigned long long bad = 0;
      GMD_HIP(hipMemcpyAsync(&bad, d.bad, sizeof bad, hipMemcpyDeviceToHost, d.stream));
      GMD_HIP(hipStreamSynchronize(d.stream));
      if (bad) *ok = false;
    }
    return GMD_OK;
  }
};

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int resolve_devices(int ndev, const int* devices, std::vector<int>* out) {
  int visible = 0;
  GMD_HIP(hipGetDeviceCount(&visible));
  if (ndev <= 0) ndev = visible;
  if (ndev > visible || ndev > 64) {
    gmd::set_error("%d GPUs requested, %d visible", ndev, visible);
    return GMD_ERR_BADPARAM;
  }
  for (int i = 0; i < ndev; i++) {
    int id = devices ? devices[i] : i;
    if (id < 0 || id >= visible || std::find(out->begin(), out->end(), id) != out->end()) {
      gmd::set_error("bad or repeated device index %d", id);
      return GMD_ERR_BADPARAM;
    }
    out->push_back(id);
  }
  return GMD_OK;
}

}  // namespace

extern "C" {

const char* gmd_collective_name(int op) {
  static const char* names[] = {"all_reduce", "all_gather", "reduce_scatter", "all_to_all", "broadcast"};
  return op >= 0 && op < GMD_COLL_COUNT ? names[op] : "unknown";
}

double gmd_busbw_factor(int op, int ndev) {
  if (ndev <= 1) return 1.0;
  switch (op) {
    case GMD_COLL_ALL_REDUCE: return 2.0 * (ndev - 1) / ndev;
    case GMD_COLL_ALL_GATHER:
    case GMD_COLL_REDUCE_SCATTER:
    case GMD_COLL_ALL_TO_ALL: return (double)(ndev - 1) / ndev;
    default: return 1.0;
  }
}

int gmd_rccl_sweep(const gmd_rccl_config_t* cfg_in, gmd_rccl_point_t* out, int* n_out) {
  if (!out || !n_out || *n_out <= 0) return GMD_ERR_BADPARAM;
  gmd_rccl_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  if (cfg_in) cfg = *cfg_in;
  if (!cfg.ops_mask) cfg.ops_mask = (1u << GMD_COLL_COUNT) - 1;
  if (!cfg.min_bytes) cfg.min_bytes = 8;
  if (!cfg.max_bytes) cfg.max_bytes = 1ull << 30;
  if (cfg.factor < 2) cfg.factor = 4;
  if (cfg.warmup <= 0) cfg.warmup = 3;
  if (cfg.min_bytes > cfg.max_bytes) return GMD_ERR_BADPARAM;
  const int cap = *n_out;
  *n_out = 0;

  Rccl* R = load_rccl();
  if (!R) return GMD_ERR_UNSUPPORTED;
  std::vector<int> ids;
  int rc = resolve_devices(cfg.ndev, cfg.devices, &ids);
  if (rc) return rc;
  const int w = (int)ids.size();

  Sweep s;
  s.R = R;
  s.devs.resize(w);
  // Payload in floats, rounded up so that it splits across the GPUs.
  const uint64_t max_elems = ((cfg.max_bytes / 4 + w - 1) / w) * w + w;
  for (int r = 0; r < w; r++) {
    Dev& d = s.devs[r];
    d.id = ids[r];
    GMD_HIP(hipSetDevice(d.id));
    GMD_HIP(hipStreamCreateWithFlags(&d.stream, hipStreamNonBlocking));
    GMD_HIP(hipEventCreate(&d.t0));
    GMD_HIP(hipEventCreate(&d.t1));
    GMD_HIP(hipMalloc((void**)&d.send, max_elems * 4));
    GMD_HIP(hipMalloc((void**)&d.recv, max_elems * 4));
    GMD_HIP(hipMalloc((void**)&d.bad, sizeof(unsigned long long)));
    GMD_HIP(hipMemsetAsync(d.send, 0, max_elems * 4, d.stream));
    GMD_HIP(hipMemsetAsync(d.recv, 0, max_elems * 4, d.stream));
  }
  s.comms.assign(w, nullptr);
  GMD_NCCL(R, R->ncclCommInitAll(s.comms.data(), w, ids.data()));
  // The library's own view of what was built: w ranks, rank r on the r-th listed device.  A communicator that came
  // up smaller, or with two ranks on one card, would still move data and produce plausible numbers.
  for (int r = 0; r < w; r++) {
    int count = 0, rank = -1, dev = -1;
    GMD_NCCL(R, R->ncclCommCount(s.comms[r], &count));
    GMD_NCCL(R, R->ncclCommUserRank(s.comms[r], &rank));
    GMD_NCCL(R, R->ncclCommCuDevice(s.comms[r], &dev));
    if (count != w || rank != r || dev != ids[r]) {
      gmd::set_error("communicator %d reports %d ranks, rank %d on device %d; expected %d ranks, rank %d on device %d", r, count,
                     rank, dev, w, r, ids[r]);
      return GMD_ERR_VERIFY;
    }
  }

  std::vector<uint64_t> sizes;
  for (uint64_t b = cfg.min_bytes; b < cfg.max_bytes; b *= cfg.factor) sizes.push_back(b);
  sizes.push_back(cfg.max_bytes);

  for (int op = 0; op < GMD_COLL_COUNT; op++) {
    if (!(cfg.ops_mask & (1u << op))) continue;
    const bool sharded = op == GMD_COLL_ALL_GATHER || op == GMD_COLL_REDUCE_SCATTER || op == GMD_COLL_ALL_TO_ALL;
    for (uint64_t bytes : sizes) {
      if (*n_out >= cap) return GMD_OK;
      uint64_t n = std::max<uint64_t>(1, bytes / 4);
      if (sharded) n = std::max<uint64_t>(w, (n / w) * w);
      gmd_rccl_point_t p;
      memset(&p, 0, sizeof p);
      p.op = op;
      p.ndev = w;
      p.bytes = n * 4;
      if (cfg.verify) {
        bool ok = false;
        if ((rc = s.verify(op, n, &ok))) return rc;
        p.verified = ok ? 1 : -1;
      }
      const int iters = cfg.iters > 0 ? cfg.iters : (p.bytes <= (1u << 20) ? 50 : p.bytes <= (64u << 20) ? 20 : 8);
      for (int i = 0; i < cfg.warmup; i++)
        if ((rc = s.launch(op, n))) return rc;
      if ((rc = s.sync_all())) return rc;
      if ((rc = s.mark(true))) return rc;
      for (int i = 0; i < iters; i++)
        if ((rc = s.launch(op, n))) return rc;
      double total = 0;
      if ((rc = s.mark(false)) || (rc = s.elapsed_s(&total)) || (rc = s.sync_all())) return rc;
      const double dt = std::max(total, 1e-9) / iters;
      p.iters = iters;
      p.time_us = dt * 1e6;
      p.algbw_gbps = p.bytes / dt / 1e9;
      p.busbw_gbps = p.algbw_gbps * gmd_busbw_factor(op, w);
      out[(*n_out)++] = p;
    }
  }
  return GMD_OK;
}

int gmd_p2p_matrix(int ndev, const int* devices, uint64_t bytes, int iters, double* gbps) {
  if (!gbps) return GMD_ERR_BADPARAM;
  std::vector<int> ids;
  int rc = resolve_devices(ndev, devices, &ids);
  if (rc) return rc;
  const int w = (int)ids.size();
  if (!bytes) bytes = 256ull << 20;
  if (iters <= 0) iters = 5;
  std::vector<gmd::DeviceBuffer> src(w), dst(w);
  struct Streams {  // destroyed on every way out, each on its own device
    std::vector<int> ids;
    std::vector<hipStream_t> s;
    ~Streams() {
      for (size_t i = 0; i < s.size(); i++) {
        if (!s[i]) continue;
        (void)hipSetDevice(ids[i]);
        (void)hipStreamDestroy(s[i]);
      }
    }
  } owned;
  owned.ids = ids;
  owned.s.assign(w, nullptr);
  std::vector<hipStream_t>& streams = owned.s;
  for (int i = 0; i < w; i++) {
    GMD_HIP(hipSetDevice(ids[i]));
    if ((rc = src[i].alloc(bytes)) || (rc = dst[i].alloc(bytes))) return rc;
    GMD_HIP(hipMemset(src[i].p, 0x5a, bytes));
    GMD_HIP(hipStreamCreateWithFlags(&streams[i], hipStreamNonBlocking));
  }
  for (int i = 0; i < w; i++) {
    GMD_HIP(hipSetDevice(ids[i]));
    for (int j = 0; j < w; j++) {
      double& cell = gbps[i * w + j];
      cell = 0;
      if (i == j) continue;
      int can = 0;
      GMD_HIP(hipDeviceCanAccessPeer(&can, ids[i], ids[j]));
      if (!can) {
        cell = -1;
        continue;
      }
      hipError_t e = hipDeviceEnablePeerAccess(ids[j], 0);
      if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled) {
        cell = -1;
        (void)hipGetLastError();
        continue;
      }
      (void)hipGetLastError();
      // The sender drives the copy (a write over the link), on its own stream.
      GMD_HIP(hipMemcpyPeerAsync(dst[j].p, ids[j], src[i].p, ids[i], bytes, streams[i]));
      GMD_HIP(hipStreamSynchronize(streams[i]));
      const double t0 = now_s();
      for (int k = 0; k < iters; k++) GMD_HIP(hipMemcpyPeerAsync(dst[j].p, ids[j], src[i].p, ids[i], bytes, streams[i]));
      GMD_HIP(hipStreamSynchronize(streams[i]));
      cell = bytes * (double)iters / (now_s() - t0) / 1e9;
    }
  }
  return GMD_OK;
}

}  // extern "C"
