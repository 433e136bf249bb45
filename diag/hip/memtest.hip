// HBM memtest for MI355X: pattern sweep over (almost) all of the 288 GB HBM3E.
//
// For each pattern the whole region is written, then read back and compared
// against the value recomputed from the word index — no reference copy, so the
// test covers twice the memory a copy-compare test could.  Patterns follow the
// classic memory-test repertoire (walking ones / zeros, address-in-address,
// checkerboard and its inverse, a counter-based pseudo-random fill, modulo-20).
// Both passes move 16 bytes per lane per access (two 64-bit words).  The write
// pass is one store per lane over a very large grid (see fill_kernel); in the read
// pass every workgroup walks its own contiguous span (128 workgroups per CU).  The
// pattern is a template parameter, so the kernels are loads / stores plus a few
// integer ops.  Mismatches are counted with one atomic per mismatching lane
// (rare) and the first few are logged with the offset in the region, the chunk, the
// device virtual address, its offset in the 4 KiB page (the granule in which
// consecutive addresses change HBM channel), the expected and the actual value and
// the number of bits that differ: the operator's starting point for mapping a fault
// to a stack or channel (the physical mapping itself is not visible to a process).
//
// Between the write and the verify pass a scrub buffer of twice the 256 MiB
// Infinity Cache is overwritten, so that the lines written last have been pushed
// out to the DRAM cells before they are read back: the cache sits in front of the
// memory, and a test that reads its own tail out of it has not tested that memory.
//
// A march pass (optional) reads every word, checks it and writes its complement in
// one kernel, up and then down again: each cell is read after having held a value
// and its inverse, and the kernel is a read+write stress at the same time.
//
// The region is allocated in chunks (default 2 GiB) so a fragmented heap still
// yields high coverage; allocation stops at the first failure and the achieved
// coverage is reported.
#include <chrono>

#include "diag_common.h"

namespace {

constexpr int kBlock = 256;
constexpr int kUnroll = 4;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// Value of 64-bit word `w` (global index inside the tested region).
__host__ __device__ __forceinline__ uint64_t expected_word(uint32_t pattern, uint64_t w, uint64_t seed) {
  switch (pattern) {
    case GMD_PAT_WALKING_ONES: return 1ull << ((w + seed) & 63);
    case GMD_PAT_WALKING_ZEROS: return ~(1ull << ((w + seed) & 63));
    case GMD_PAT_ADDRESS: return w * 8;  // its own byte offset
    case GMD_PAT_CHECKERBOARD: return (w & 1) ? 0x5555555555555555ull : 0xAAAAAAAAAAAAAAAAull;
    case GMD_PAT_INV_CHECKERBOARD: return (w & 1) ? 0xAAAAAAAAAAAAAAAAull : 0x5555555555555555ull;
    case GMD_PAT_RANDOM: return splitmix64(w ^ (seed * 0xD1342543DE82EF95ull));
    case GMD_PAT_MODULO_20: return (w % 20 == seed % 20) ? 0xFFFFFFFFFFFFFFFFull : 0ull;
  }
  return 0;
}

constexpr unsigned kVerifyMaxGrid = (1u << 24) - 1;

// PAT is a compile-time constant so the pattern switch folds away; every workgroup
// walks its own contiguous span (see gmd::span_plan).
// The write pass gives every lane ONE 16-byte store: workgroup w fills elements
// [256 w, 256 w + 256) and exits, and the in-order dispatch of the (very large)
// grid does the traversal, so the stores in flight cover one compact window that
// slides through the chunk.  Measured against 128 KiB spans per workgroup this is
// 6.0 -> 6.85 TB/s, and 7.1 TB/s with 4 workgroups per CU instead of 8 (kFillLdsBytes
// of unused dynamic LDS is the occupancy cap).  Plain stores: a pure store stream
// is faster without the non-temporal hint.
constexpr int kFillLdsBytes = 40 * 1024;
template <uint32_t PAT>
__global__ __launch_bounds__(kBlock) void fill_kernel(u64x2* __restrict__ buf, uint64_t n2, uint64_t base_word, uint64_t seed) {
  const uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
  if (i >= n2) return;
  const uint64_t w = base_word + 2 * i;
  buf[i] = u64x2{expected_word(PAT, w, seed), expected_word(PAT, w + 1, seed)};
}

struct VerifyOut {
  unsigned long long errors;
  unsigned int logged;
  unsigned int pad;
  gmd_memtest_error_t log[GMD_MEMTEST_LOG];
};

__device__ __forceinline__ void report(VerifyOut* out, uint64_t word, uint64_t expect, uint64_t got, uint32_t pattern, const void* where) {
  atomicAdd(&out->errors, 1ull);
  unsigned slot = atomicAdd(&out->logged, 1u);
  if (slot < GMD_MEMTEST_LOG) {
    out->log[slot].byte_offset = word * 8;
    out->log[slot].expected = expect;
    out->log[slot].actual = got;
    out->log[slot].pattern = pattern;
    out->log[slot].chunk = 0;
    out->log[slot].device_address = (uint64_t)where;
    out->log[slot].flipped_bits = (uint32_t)__popcll(expect ^ got);
    out->log[slot].page_offset = (uint32_t)((uint64_t)where & 4095);
  }
}

// INV: the region holds the complement of the pattern (second half of a march).
template <uint32_t PAT, bool INV>
__device__ __forceinline__ uint64_t want_word(uint64_t w, uint64_t seed) {
  const uint64_t e = expected_word(PAT, w, seed);
  return INV ? ~e : e;
}

template <uint32_t PAT, bool INV>
__global__ __launch_bounds__(kBlock) void verify_kernel(const u64x2* __restrict__ buf, uint64_t n2, uint64_t span,
                                                        uint64_t base_word, uint64_t seed, VerifyOut* out) {
  uint64_t i = (uint64_t)blockIdx.x * span + threadIdx.x;
  const uint64_t end = min((uint64_t)(blockIdx.x + 1) * span, n2);
  for (; i + (kUnroll - 1) * kBlock < end; i += kUnroll * kBlock) {
    u64x2 v[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; u++) v[u] = __builtin_nontemporal_load(buf + i + u * kBlock);
    // One cheap combined test per pass; the per-word reporting runs only when it trips.
    uint64_t diff = 0;
#pragma unroll
    for (int u = 0; u < kUnroll; u++) {
      const uint64_t w = base_word + 2 * (i + u * kBlock);
      diff |= (v[u].x ^ want_word<PAT, INV>(w, seed)) | (v[u].y ^ want_word<PAT, INV>(w + 1, seed));
    }
    if (diff) {
#pragma unroll
      for (int u = 0; u < kUnroll; u++) {
        const uint64_t w = base_word + 2 * (i + u * kBlock);
        const uint64_t e0 = want_word<PAT, INV>(w, seed), e1 = want_word<PAT, INV>(w + 1, seed);
        if (v[u].x != e0) report(out, w, e0, v[u].x, PAT, (const uint64_t*)(buf + i + u * kBlock));
        if (v[u].y != e1) report(out, w + 1, e1, v[u].y, PAT, (const uint64_t*)(buf + i + u * kBlock) + 1);
      }
    }
  }
  for (; i < end; i += kBlock) {
    u64x2 v = __builtin_nontemporal_load(buf + i);
    const uint64_t w = base_word + 2 * i;
    const uint64_t e0 = want_word<PAT, INV>(w, seed), e1 = want_word<PAT, INV>(w + 1, seed);
    if (v.x != e0) report(out, w, e0, v.x, PAT, (const uint64_t*)(buf + i));
    if (v.y != e1) report(out, w + 1, e1, v.y, PAT, (const uint64_t*)(buf + i) + 1);
  }
}

// March element: read a word, check it against what the previous element left, write its complement.  One access
// pair per lane over an in-order grid like the write pass; DOWN walks the region from its end (the classic march
// goes through the addresses in both directions).  16 bytes read and 16 bytes written per lane.
template <uint32_t PAT, bool INV, bool DOWN>
__global__ __launch_bounds__(kBlock) void march_kernel(u64x2* __restrict__ buf, uint64_t n2, uint64_t base_word, uint64_t seed, VerifyOut* out) {
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n2) return;
  if (DOWN) i = n2 - 1 - i;
  const u64x2 v = __builtin_nontemporal_load(buf + i);
  const uint64_t w = base_word + 2 * i;
  const uint64_t e0 = want_word<PAT, INV>(w, seed), e1 = want_word<PAT, INV>(w + 1, seed);
  if (v.x != e0) report(out, w, e0, v.x, PAT, (const uint64_t*)(buf + i));
  if (v.y != e1) report(out, w + 1, e1, v.y, PAT, (const uint64_t*)(buf + i) + 1);
  __builtin_nontemporal_store(u64x2{~e0, ~e1}, buf + i);  // 6.05 -> 6.7 TB/s over a cached store: the line is not read again
}

// Host-side dispatch from the run-time pattern bit to the compile-time instance.
#define GMD_FOR_PATTERN(pattern, CALL)                                   \
  switch (pattern) {                                                     \
    case GMD_PAT_WALKING_ONES: CALL(GMD_PAT_WALKING_ONES); break;        \
    case GMD_PAT_WALKING_ZEROS: CALL(GMD_PAT_WALKING_ZEROS); break;      \
    case GMD_PAT_ADDRESS: CALL(GMD_PAT_ADDRESS); break;                  \
    case GMD_PAT_CHECKERBOARD: CALL(GMD_PAT_CHECKERBOARD); break;        \
    case GMD_PAT_INV_CHECKERBOARD: CALL(GMD_PAT_INV_CHECKERBOARD); break; \
    case GMD_PAT_RANDOM: CALL(GMD_PAT_RANDOM); break;                    \
    case GMD_PAT_MODULO_20: CALL(GMD_PAT_MODULO_20); break;              \
  }

void launch_fill(u64x2* buf, uint64_t n2, uint64_t base_word, uint32_t pattern, uint64_t seed, hipStream_t st) {
  // A launch holds fewer than 2^32 work-items: 32 GiB per launch.
  constexpr uint64_t kPerLaunch = 1ull << 31;
  for (uint64_t off = 0; off < n2; off += kPerLaunch) {
    const uint64_t n = std::min(kPerLaunch, n2 - off);
    const unsigned grid = (unsigned)((n + kBlock - 1) / kBlock);
#define CALL(P) fill_kernel<P><<<grid, kBlock, kFillLdsBytes, st>>>(buf + off, n, base_word + 2 * off, seed)
    GMD_FOR_PATTERN(pattern, CALL)
#undef CALL
  }
}

void launch_verify(const u64x2* buf, uint64_t n2, uint64_t base_word, uint32_t pattern, uint64_t seed, VerifyOut* out,
                   hipStream_t st) {
  // One unrolled pass (16 KiB) per workgroup up to 256 GiB per launch (grid x 256 work-items stays under
  // 2^32), longer spans beyond: like the write pass, the in-order grid keeps the loads in flight in one
  // compact window (+3 % over 128 workgroups per CU).
  unsigned grid;
  uint64_t span;
  gmd::span_plan(n2, (uint64_t)kBlock * kUnroll, kVerifyMaxGrid, &grid, &span);
#define CALL(P) verify_kernel<P, false><<<grid, kBlock, 0, st>>>(buf, n2, span, base_word, seed, out)
  GMD_FOR_PATTERN(pattern, CALL)
#undef CALL
}

// One march element over a chunk: the region holds the pattern (inv = false) or its complement, and is left
// holding the other.
void launch_march(u64x2* buf, uint64_t n2, uint64_t base_word, uint32_t pattern, uint64_t seed, bool inv, bool down, VerifyOut* out,
                  hipStream_t st) {
  constexpr uint64_t kPerLaunch = 1ull << 31;
  // whole launches in address order going up, in reverse order going down
  const uint64_t launches = (n2 + kPerLaunch - 1) / kPerLaunch;
  for (uint64_t k = 0; k < launches; k++) {
    const uint64_t off = (down ? launches - 1 - k : k) * kPerLaunch;
    const uint64_t n = std::min(kPerLaunch, n2 - off);
    // One wave per workgroup and no cap on workgroups per CU, the shape of the copy stream: a kernel that reads and
    // writes wants many small workgroups in flight (+14 % over the pure store's shape of 256 lanes, four per CU).
    constexpr unsigned blk = 64, lds = 0;
    const unsigned grid = (unsigned)((n + blk - 1) / blk);
#define CALL(P)                                                                                                          \
  do {                                                                                                                   \
    if (!inv && !down) march_kernel<P, false, false><<<grid, blk, lds, st>>>(buf + off, n, base_word + 2 * off, seed, out); \
    else if (inv && down) march_kernel<P, true, true><<<grid, blk, lds, st>>>(buf + off, n, base_word + 2 * off, seed, out); \
    else if (inv) march_kernel<P, true, false><<<grid, blk, lds, st>>>(buf + off, n, base_word + 2 * off, seed, out);       \
    else march_kernel<P, false, true><<<grid, blk, lds, st>>>(buf + off, n, base_word + 2 * off, seed, out);                \
  } while (0)
    GMD_FOR_PATTERN(pattern, CALL)
#undef CALL
  }
}

__global__ void xor_word_kernel(unsigned long long* p, unsigned long long mask) { *p ^= mask; }

double now_s() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

bool valid_pattern(uint32_t p) { return p && !(p & (p - 1)) && (p & GMD_PAT_ALL); }

}  // namespace

extern "C" {

int gmd_memtest_fill(void* buf, uint64_t n_words, uint64_t base_word, uint32_t pattern, uint64_t seed, void* hip_stream) {
  if (!buf || !n_words || (n_words & 1) || ((uintptr_t)buf & 15) || !valid_pattern(pattern)) return GMD_ERR_BADPARAM;
  const uint64_t n2 = n_words / 2;
  launch_fill((u64x2*)buf, n2, base_word, pattern, seed, (hipStream_t)hip_stream);
  GMD_HIP(hipGetLastError());
  return GMD_OK;
}

int gmd_memtest_verify(const void* buf, uint64_t n_words, uint64_t base_word, uint32_t pattern, uint64_t seed,
                       uint64_t* errors_out, gmd_memtest_error_t* log, uint32_t* logged, void* hip_stream) {
  if (!buf || !n_words || (n_words & 1) || ((uintptr_t)buf & 15) || !valid_pattern(pattern) || !errors_out)
    return GMD_ERR_BADPARAM;
  hipStream_t st = (hipStream_t)hip_stream;
  VerifyOut* d = nullptr;
  GMD_HIP(hipMalloc((void**)&d, sizeof(VerifyOut)));
  hipError_t e = hipMemsetAsync(d, 0, sizeof(VerifyOut), st);
  const uint64_t n2 = n_words / 2;
  if (e == hipSuccess) {
    launch_verify((const u64x2*)buf, n2, base_word, pattern, seed, d, st);
    e = hipGetLastError();
  }
  VerifyOut h;
  if (e == hipSuccess) e = hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  (void)hipFree(d);
  if (e != hipSuccess) {
    gmd::set_error("memtest verify failed: %s", hipGetErrorString(e));
    return GMD_ERR_HIP;
  }
  *errors_out = h.errors;
  uint32_t n = h.logged < GMD_MEMTEST_LOG ? h.logged : GMD_MEMTEST_LOG;
  if (logged) *logged = n;
  if (log) memcpy(log, h.log, n * sizeof(gmd_memtest_error_t));
  return GMD_OK;
}

int gmd_memtest(int device, const gmd_memtest_config_t* cfg_in, gmd_memtest_result_t* out) {
  if (!out) return GMD_ERR_BADPARAM;
  memset(out, 0, sizeof *out);
  gmd_memtest_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  if (cfg_in) cfg = *cfg_in;
  if (cfg.fraction <= 0 || cfg.fraction > 0.98) cfg.fraction = 0.90;
  if (!cfg.chunk_bytes) cfg.chunk_bytes = 2ull << 30;
  cfg.chunk_bytes &= ~(uint64_t)4095;
  if (cfg.chunk_bytes < 4096) return GMD_ERR_BADPARAM;
  if (!cfg.patterns) cfg.patterns = GMD_PAT_ALL;
  if (!cfg.passes) cfg.passes = 1;
  GMD_HIP(hipSetDevice(device));
  size_t free_b = 0, total_b = 0;
  GMD_HIP(hipMemGetInfo(&free_b, &total_b));
  uint64_t want = cfg.bytes ? cfg.bytes : (uint64_t)((double)free_b * cfg.fraction);
  want &= ~(uint64_t)4095;
  if (want < 4096) return GMD_ERR_BADPARAM;
  out->bytes_requested = want;

  struct Chunk {
    void* p;
    uint64_t bytes, base_word;
  };
  std::vector<Chunk> chunks;
  uint64_t got = 0;
  while (got < want) {
    uint64_t sz = std::min<uint64_t>(cfg.chunk_bytes, want - got);
    void* p = nullptr;
    if (hipMalloc(&p, sz) != hipSuccess) {
      (void)hipGetLastError();  // fragmentation or a neighbour's allocation: test what we have
      break;
    }
    chunks.push_back({p, sz, got / 8});
    got += sz;
  }
  auto cleanup = [&]() {
    for (auto& c : chunks) (void)hipFree(c.p);
  };
  if (chunks.empty()) {
    gmd::set_error("memtest: could not allocate any device memory (%llu bytes wanted)", (unsigned long long)want);
    return GMD_ERR_NOMEM;
  }
  out->bytes_tested = got;
  out->chunks = (uint32_t)chunks.size();

  VerifyOut* d_out = nullptr;
  if (hipMalloc((void**)&d_out, sizeof(VerifyOut)) != hipSuccess) {
    cleanup();
    gmd::set_error("memtest: out of memory for the result block");
    return GMD_ERR_NOMEM;
  }
  // Twice the Infinity Cache of scrub space; without it (memory all taken) the test still runs, and says so.
  constexpr uint64_t kScrubBytes = 512ull << 20;
  void* scrub = nullptr;
  if (hipMalloc(&scrub, kScrubBytes) != hipSuccess) {
    (void)hipGetLastError();
    scrub = nullptr;
  }
  out->scrubbed = scrub ? 1 : 0;
  auto push_out = [&](uint64_t seed) {  // what the last pass left in the cache goes out to the memory cells
    if (scrub) launch_fill((u64x2*)scrub, kScrubBytes / 16, 0, GMD_PAT_RANDOM, seed, 0);
  };
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const double t_start = now_s();
  double write_tbps_sum = 0, verify_tbps_sum = 0;
  int npat = 0;
  for (uint32_t b = 0; b < 7; b++)
    if (cfg.patterns & (1u << b)) npat++;
  const int total_steps = npat * (int)cfg.passes * 2 + (cfg.march ? 4 * (int)cfg.passes : 0);
  int step = 0;
  hipError_t err = hipSuccess;
  for (uint32_t pass = 0; pass < cfg.passes && err == hipSuccess; pass++) {
    for (uint32_t b = 0; b < 7 && err == hipSuccess; b++) {
      const uint32_t pat = 1u << b;
      if (!(cfg.patterns & pat)) continue;
      const uint64_t seed = cfg.seed + pass * 7919 + b;
      // ---- write pass
      float ms = 0;
      (void)hipEventRecord(e0, 0);
      for (auto& c : chunks) {
        const uint64_t n2 = c.bytes / 16;
        launch_fill((u64x2*)c.p, n2, c.base_word, pat, seed, 0);
      }
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      if (err != hipSuccess) break;
      (void)hipEventElapsedTime(&ms, e0, e1);
      write_tbps_sum += (double)got / (ms * 1e-3) / 1e12;
      if (cfg.progress) cfg.progress((double)++step / total_steps, "write", cfg.progress_user);
      // ---- optional fault injection
      if (cfg.inject_xor && cfg.inject_offset / 8 < got / 8) {
        const uint64_t word = cfg.inject_offset / 8;
        for (auto& c : chunks)
          if (word >= c.base_word && word < c.base_word + c.bytes / 8)
            xor_word_kernel<<<1, 1>>>((unsigned long long*)c.p + (word - c.base_word), cfg.inject_xor);
      }
      // ---- verify pass
      push_out(seed);
      err = hipMemset(d_out, 0, sizeof(VerifyOut));
      if (err != hipSuccess) break;
      (void)hipEventRecord(e0, 0);
      for (auto& c : chunks) {
        const uint64_t n2 = c.bytes / 16;
        launch_verify((const u64x2*)c.p, n2, c.base_word, pat, seed, d_out, 0);
      }
      (void)hipEventRecord(e1, 0);
      err = hipEventSynchronize(e1);
      if (err != hipSuccess) break;
      (void)hipEventElapsedTime(&ms, e0, e1);
      verify_tbps_sum += (double)got / (ms * 1e-3) / 1e12;
      VerifyOut h;
      err = hipMemcpy(&h, d_out, sizeof h, hipMemcpyDeviceToHost);
      if (err != hipSuccess) break;
      out->errors += h.errors;
      for (uint32_t i = 0; i < h.logged && i < GMD_MEMTEST_LOG && out->logged < GMD_MEMTEST_LOG; i++) {
        gmd_memtest_error_t e = h.log[i];
        for (size_t ci = 0; ci < chunks.size(); ci++)
          if (e.byte_offset / 8 >= chunks[ci].base_word && e.byte_offset / 8 < chunks[ci].base_word + chunks[ci].bytes / 8)
            e.chunk = (uint32_t)ci;
        out->log[out->logged++] = e;
      }
      out->patterns_run++;
      if (cfg.progress) cfg.progress((double)++step / total_steps, "verify", cfg.progress_user);
    }
    // ---- march: write the random pattern, then up (check, write complement), down (check complement, write the
    //      pattern), and a last check.  Every cell has been read after holding a value and its complement.
    if (cfg.march && err == hipSuccess) {
      const uint32_t pat = GMD_PAT_RANDOM;
      const uint64_t seed = cfg.seed + pass * 7919 + 101;
      for (auto& c : chunks) launch_fill((u64x2*)c.p, c.bytes / 16, c.base_word, pat, seed, 0);
      if (cfg.inject_xor && cfg.inject_offset / 8 < got / 8) {
        const uint64_t word = cfg.inject_offset / 8;
        for (auto& c : chunks)
          if (word >= c.base_word && word < c.base_word + c.bytes / 8)
            xor_word_kernel<<<1, 1>>>((unsigned long long*)c.p + (word - c.base_word), cfg.inject_xor);
      }
      if (cfg.progress) cfg.progress((double)++step / total_steps, "march: write", cfg.progress_user);
      err = hipMemset(d_out, 0, sizeof(VerifyOut));
      double march_bytes = 0, march_s = 0;
      for (int element = 0; element < 2 && err == hipSuccess; element++) {
        const bool down = element == 1;
        push_out(seed + element);
        float ms = 0;
        (void)hipEventRecord(e0, 0);
        for (size_t k = 0; k < chunks.size(); k++) {
          auto& c = chunks[down ? chunks.size() - 1 - k : k];
          launch_march((u64x2*)c.p, c.bytes / 16, c.base_word, pat, seed, /*inv=*/down, down, d_out, 0);
        }
        (void)hipEventRecord(e1, 0);
        err = hipEventSynchronize(e1);
        if (err != hipSuccess) break;
        (void)hipEventElapsedTime(&ms, e0, e1);
        march_bytes += 2.0 * (double)got;
        march_s += ms * 1e-3;
        if (cfg.progress) cfg.progress((double)++step / total_steps, down ? "march: down" : "march: up", cfg.progress_user);
      }
      if (err == hipSuccess) {
        push_out(seed + 2);
        for (auto& c : chunks) launch_verify((const u64x2*)c.p, c.bytes / 16, c.base_word, pat, seed, d_out, 0);
        VerifyOut h;
        err = hipMemcpy(&h, d_out, sizeof h, hipMemcpyDeviceToHost);
        if (err == hipSuccess) {
          out->errors += h.errors;
          for (uint32_t i = 0; i < h.logged && i < GMD_MEMTEST_LOG && out->logged < GMD_MEMTEST_LOG; i++) {
            gmd_memtest_error_t e = h.log[i];
            for (size_t ci = 0; ci < chunks.size(); ci++)
              if (e.byte_offset / 8 >= chunks[ci].base_word && e.byte_offset / 8 < chunks[ci].base_word + chunks[ci].bytes / 8)
                e.chunk = (uint32_t)ci;
            out->log[out->logged++] = e;
          }
          out->march_passes++;
          if (march_s > 0) out->march_tbps = march_bytes / march_s / 1e12;
        }
        if (cfg.progress) cfg.progress((double)++step / total_steps, "march: check", cfg.progress_user);
      }
    }
  }
  if (err == hipSuccess) err = hipGetLastError();
  out->seconds = now_s() - t_start;
  if (out->patterns_run) {
    out->write_tbps = write_tbps_sum / out->patterns_run;
    out->verify_tbps = verify_tbps_sum / out->patterns_run;
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(d_out);
  if (scrub) (void)hipFree(scrub);
  cleanup();
  if (err != hipSuccess) {
    gmd::set_error("memtest failed: %s", hipGetErrorString(err));
    return GMD_ERR_HIP;
  }
  return GMD_OK;
}

}  // extern "C"
