// MFMA targeted-stress diagnostics for MI355X (gfx950).
//
// (1) Matrix-pipe peak: every wave issues back-to-back MFMAs on register
//     operands (no memory traffic) — the rate the matrix cores sustain at the
//     clock the chip holds under that load.  bf16, fp8 (same rate as bf16),
//     block-scaled MX-fp8 (K=128, 2x bf16 per clock) and f32-input forms.
//
// (2) GEMM stress: C[M,N] = A[M,K] * B[N,K]^T, 256x256 output tile per
//     512-thread workgroup (8 waves as 2(M) x 4(N), 128x64 per wave, 32
//     accumulators of mfma_f32_16x16x32), K-step of 128 bytes per row
//     (64 bf16 / 128 fp8), operands staged HBM -> LDS by direct 16-byte
//     LDS-DMA (global_load_lds) into two 64 KiB buffers and read back with
//     ds_read_b128.
//
//     LDS image (both operands): 8-row x 128-byte pieces of 1 KiB — exactly
//     what one LDS-DMA wave-instruction writes, so every instruction fetches
//     eight whole 128-byte lines — with the 16-byte chunks of each row
//     XOR-permuted:  chunk' = chunk ^ f(row & 7).  ds_read_b128 is serviced in
//     four 16-lane groups and each group of an MFMA fragment read touches all
//     16 rows once with two chunk values, so f is chosen to make the 16 slots
//     of every group distinct: f = 2*((row&7)>>1) (a group's lanes read chunks
//     {c, c^1} over its 16 rows).
//     Conflict-free reads, fully coalesced fetches, linear DMA destination;
//     the permutation lives in the per-lane SOURCE address and in the read
//     address, never in the DMA destination.
//
//     The MFMA takes the B fragment as its first operand and the A fragment as
//     its second, which transposes the accumulator tile: each lane ends up with
//     four consecutive N elements of one C row and stores them as one 8-byte
//     word instead of four 2-byte scatters.
//
//     Workgroup ids are remapped so the tiles one XCD works on are contiguous
//     (its private L2 then holds a few A row-panels and the B column panels it
//     re-reads) with the bijective form that is valid for any grid size.
//
//     Data types.  bf16: mfma_f32_16x16x32_bf16, one per 16-byte fragment read.
//     fp8 (OCP e4m3, unscaled): mfma_f32_16x16x32_fp8_fp8, two per 16-byte read —
//     the low and the high 8 bytes; which k a lane's bytes stand for only has to be
//     the same in both operands, so the bf16 image and swizzle serve unchanged.
//     MX-fp8: mfma_scale_f32_16x16x128_f8f6f4 with one E8M0 scale per row and 32
//     consecutive k (the OCP MX block).  Lane (fr, fq) holds k 16*fq .. +15 in operand
//     registers 0-3 and k 64 + 16*fq .. +15 in registers 4-7 — the same two 16-byte
//     chunks a bf16 lane reads — and supplies the scale of row fr, block fq (k 32*fq ..
//     +31; measured with one-hot scales and operands, scripts/gpu_mx_probe.py).
//     Scales arrive "packed": per 128 rows, per half h and per K-tile one 256-byte
//     unit whose dword fq*16 + fr holds, in byte c, the scale of row 8*fr + c + 4*h
//     of the 128, block fq: the rows that fragment lane fr works on in the four
//     16-row blocks c of a half-tile (rows are dealt to lanes in groups of eight, see
//     gemm_w4_kernel).  A wave reads the unit with one dword per lane and selects the
//     byte of the block with the instruction's op_sel field: no shifts, no masks.
//     gmd_mx_pack_scales builds the packed form from a row-major [rows][K/32] one.
#include <hip/hip_bf16.h>

#include <chrono>
#include <cmath>
#include <cstdlib>
#include <algorithm>
#include <mutex>
#include <type_traits>
#include <vector>

#include "diag_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef long i64x1;

// One unscaled fp8 MFMA on the low (H = 0) or high (H = 1) 8 bytes of a 16-byte fragment read.
template <int H>
__device__ __forceinline__ f32x4 mfma_fp8_half(const i32x4& b, const i32x4& a, const f32x4& c) {
  const long bl = ((long)(unsigned)b[2 * H + 1] << 32) | (unsigned)b[2 * H];
  const long al = ((long)(unsigned)a[2 * H + 1] << 32) | (unsigned)a[2 * H];
  return __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(bl, al, c, 0, 0, 0);
}
// Packed MX scale units: 64 dwords per (64-row group, K-tile).
constexpr int kScaleUnitDwords = 64;

constexpr int BM = 256, BN = 256, BKB = 128;  // K-step in BYTES per row
constexpr int kThreads = 512;
constexpr int kTileBytes = BM * BKB;           // 32 KiB per operand per buffer
constexpr int kBufBytes = 2 * kTileBytes;      // A + B
constexpr int kLdsBytes = 2 * kBufBytes;       // double buffered: 128 KiB

template <int DT>
__device__ __forceinline__ int chunk_swizzle(int row_in_piece) {
  const int q = row_in_piece >> 1;  // 0..3
  return q << 1;  // the same for every data type: all of them read chunk pairs {c, c + 4} (see the head of the file)
}

// Byte offset of 16-byte chunk `ch` (0..7) of tile row `row` inside one operand tile.
template <int DT>
__device__ __forceinline__ int lds_off(int row, int ch) {
  return (row >> 3) * 1024 + (row & 7) * 128 + ((ch ^ chunk_swizzle<DT>(row & 7)) << 4);
}

// Stage one K-step (128 bytes per row) of a 256-row operand tile into LDS.
// 32 pieces of 8 rows; wave w issues pieces 4w..4w+3.  Lane i of a piece lands at
// piece_base + 16*i, so it fetches row (i>>3), source chunk (i&7) ^ f(i>>3).
template <int DT>
__device__ __forceinline__ void stage(const char* __restrict__ g, size_t row_bytes, int kbyte, char* lds_tile, int wave,
                                      int lane) {
  const int r8 = lane >> 3;
  const int src_chunk = (lane & 7) ^ chunk_swizzle<DT>(r8);
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int piece = wave * 4 + i;
    // MX-fp8: LDS row l = 64 h' + 16 c + f (h' = l >> 6) holds tile row 128 (l >> 7) + 8 f + c + 4 (h' & 1), the order
    // the packed scales are in; the other data types stage the rows as they come.
    const int l = piece * 8 + r8;
    const int row = DT == GMD_DT_MXFP8 ? 128 * (l >> 7) + 8 * (l & 15) + ((l >> 4) & 3) + 4 * ((l >> 6) & 1) : l;
    const char* src = g + (size_t)row * row_bytes + kbyte + src_chunk * 16;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + piece * 1024), 16, 0, 0);
  }
}

// Store a wave's accumulators as bf16.  Each lane holds four consecutive N elements (8 bytes) of one row per
// 16 x 16 block; EPI = 0 stores them as they are (32 stores of 8 bytes per lane).  EPI = 1 first exchanges
// halves between neighbouring 16-lane rows of two adjacent blocks (v_permlane16_swap: odd rows of the first
// operand <-> even rows of the second), after which every lane holds eight consecutive elements of one block and
// stores 16 bytes: half the store instructions for the same bytes and addresses.  EPI = 1 also marks the stores
// non-temporal (C is never read back); EPI = 2 leaves them cached.
// PACE: one row of blocks at a time (a scheduling fence after each), for callers that hold many live registers
// across the epilogue: without it the scheduler reads all accumulators first and spills.
template <int NBLK, int EPI, bool PACE = false>
__device__ __forceinline__ void store_acc(const f32x4 (&acc)[8][NBLK], __bf16* __restrict__ C, size_t ldc, int m0, int nbase,
                                          int fq) {
  if constexpr (EPI == 0) {
#pragma unroll
    for (int mb = 0; mb < 8; mb++)
#pragma unroll
      for (int nb = 0; nb < NBLK; nb++) {
        bf16x4 o = __builtin_convertvector(acc[mb][nb], bf16x4);
        *(bf16x4*)(C + (size_t)(m0 + mb * 16) * ldc + nbase + fq * 4 + nb * 16) = o;
      }
  } else {
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int col = nbase + (fq & 1) * 16 + (fq >> 1) * 8;
#pragma unroll
    for (int mb = 0; mb < 8; mb++)
#pragma unroll
      for (int nb = 0; nb < NBLK; nb += 2) {
        u32x2 x = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[mb][nb], bf16x4));
        u32x2 y = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[mb][nb + 1], bf16x4));
        const u32x2 r0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
        const u32x2 r1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
        const u32x4 o = u32x4{r0.x, r1.x, r0.y, r1.y};
        u32x4* dst = (u32x4*)(C + (size_t)(m0 + mb * 16) * ldc + col + nb * 16);
        if constexpr (EPI == 1) __builtin_nontemporal_store(o, dst);
        else *dst = o;
        if constexpr (PACE) {
          if (nb + 2 >= NBLK) __builtin_amdgcn_sched_barrier(0);
        }
      }
  }
}

// Store a wave's 128 x 128 accumulators as bf16 when the operand rows were dealt out so that the MFMA's lane index
// runs along C's contiguous dimension (see gemm_w4_kernel): block (mb, nb) holds, in lane (fr, fq) and register r, the
// element of tile row (4 fq + r) * 8 + mb and column fr * 8 + nb.  A lane's eight nb blocks are eight consecutive
// columns, 16 bytes, and the sixteen lanes of one fq are one whole 256-byte row of the wave's tile: every quad of
// lanes writes 64 contiguous bytes.  With store_acc's layout (neighbouring lanes = neighbouring ROWS) the address
// path takes one lane per cycle: 7.9 k cycles for the 32 stores of four waves, cached or not, against 2 k at 64 B/clk.
// EPI = 3 (the soak's instantiation): EPI = 1, and the words on their way out are summed on the side: per lane, then
// over the wave, then one atomic per wave into g_gemm_reg_sum.  Identical operands give an identical sum launch after
// launch, so the host can hold EVERY launch of a soak against the first without a pass over C in memory; what that
// cannot see, a store or a memory cell that corrupts, the checksum pass over C (every Nth launch) still covers.
// The sum lives in kRegSumSlots words, one per 64-byte line, and a wave adds to the slot of its workgroup: atomics to ONE
// address are served one after the other, and the 2048 waves of a reduce kernel, all done at the same moment, took 20 us
// over theirs (2048 x 2048 x 16384 split in four: 128 -> 108 us per launch in the soak).
constexpr int kRegSumSlots = 64;
__device__ unsigned long long g_gemm_reg_sum[kRegSumSlots * 8];
__device__ __forceinline__ unsigned long long* reg_sum_slot() { return &g_gemm_reg_sum[(blockIdx.x % kRegSumSlots) * 8]; }

// One atomic per wave into g_gemm_reg_sum (the kernels that add the ranges of a split GEMM up; store_acc_rows has its own).
__device__ __forceinline__ void reg_sum_add(unsigned sum, int lane) {
  unsigned long long wide = sum;
#pragma unroll
  for (int off = 32; off; off >>= 1) wide += __shfl_xor(wide, off, 64);
  if (lane == 0) atomicAdd(reg_sum_slot(), wide);
}

template <int EPI, bool PACE = false>
__device__ __forceinline__ void store_acc_rows(const f32x4 (&acc)[8][8], __bf16* __restrict__ C, size_t ldc, int row0, int col0,
                                               int fr, int fq) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  __bf16* const base = C + (size_t)(row0 + fq * 32) * ldc + col0 + fr * 8;
  unsigned sum = 0;
#pragma unroll
  for (int mb = 0; mb < 8; mb++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      u32x4 o;
#pragma unroll
      for (int h = 0; h < 4; h++)
        o[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{acc[mb][2 * h][r], acc[mb][2 * h + 1][r]}, bf16x2));
      u32x4* dst = (u32x4*)(base + (size_t)(r * 8 + mb) * ldc);
      if constexpr (EPI == 1 || EPI == 3) __builtin_nontemporal_store(o, dst);
      else *dst = o;
      if constexpr (EPI == 3) sum += (o[0] + o[1]) + (o[2] + o[3]);
    }
    if constexpr (PACE) __builtin_amdgcn_sched_barrier(0);
  }
  if constexpr (EPI == 3) {
    unsigned long long wide = sum;
#pragma unroll
    for (int off = 32; off; off >>= 1) wide += __shfl_xor(wide, off, 64);
    if (fr == 0 && fq == 0) atomicAdd(reg_sum_slot(), wide);
  }
}

// Test hook of the soak: one bit of an operand, flipped and flipped back around a single launch.
__global__ void flip_bits_kernel(unsigned* p, unsigned mask) { *p ^= mask; }

// Moves the soak's running sum to where the host reads it (pinned host memory) and starts the next batch from zero.
// One wave, a slot per lane.
__global__ void reg_sum_harvest_kernel(unsigned long long* dst) {
  static_assert(kRegSumSlots == 64, "one slot per lane");
  unsigned long long sum = g_gemm_reg_sum[threadIdx.x * 8];
  g_gemm_reg_sum[threadIdx.x * 8] = 0;
#pragma unroll
  for (int off = 32; off; off >>= 1) sum += __shfl_xor(sum, off, 64);
  if (threadIdx.x == 0) {
    *dst = sum;
    __threadfence_system();
  }
}

// store_acc_rows for a tile that reaches beyond M or N: the same stores, each behind its own bounds test.
template <int EPI>
__device__ __forceinline__ void store_acc_rows_edge(const f32x4 (&acc)[8][8], __bf16* __restrict__ C, size_t ldc, int row0, int col0,
                                                    int fr, int fq, int M, int N) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const int col = col0 + fr * 8;
  unsigned sum = 0;  // EPI = 3: of the words stored, as in store_acc_rows
#pragma unroll
  for (int mb = 0; mb < 8; mb++) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = row0 + fq * 32 + r * 8 + mb;
      u32x4 o;
#pragma unroll
      for (int h = 0; h < 4; h++)
        o[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{acc[mb][2 * h][r], acc[mb][2 * h + 1][r]}, bf16x2));
      if (row < M && col < N) {  // N is a multiple of 8: a lane's eight columns are inside or outside together
        u32x4* dst = (u32x4*)(C + (size_t)row * ldc + col);
        if constexpr (EPI == 1 || EPI == 3) __builtin_nontemporal_store(o, dst);
        else *dst = o;
        if constexpr (EPI == 3) sum += (o[0] + o[1]) + (o[2] + o[3]);
      }
    }
  }
  if constexpr (EPI == 3) reg_sum_add(sum, fq * 16 + fr);
}

// Which range of K and which tile a workgroup of a split GEMM takes (grid = tiles x splits).  The dispatcher deals
// workgroups to the eight XCDs in turn.  Where the numbers divide, an XCD gets ONE range of K and a run of consecutive
// tiles of the grouped walk (for 64 tiles in 4 ranges: XCD x takes range x % 4 of tiles 32 (x / 4) ...), so that its 32
// workgroups share operand panels in its L2; with every range of a tile on the same XCD an XCD would fetch `splits`
// times as many panels per step (2048 x 2048 x 16384 in 4: 105 -> 75 us).  *tile is then the walk's index itself (true),
// otherwise the block id to be spread over the XCDs as in an unsplit launch (false).  Ranges under min_per K-tiles stay
// with that.  Small tiles, min_per = 32 (4 KiB of a row): an XCD that reads one short window of every operand row was
// measured slower (1024 x 1024 x 4096 in 4 ranges of 16: 19.5 -> 21.0 us; ranges of 32 and more: level or ahead,
// 33.4 -> 29.7 us at 1024 x 512 x 16384 in 8).  Large tiles, min_per = 8: never slower, 39.6 -> 36.9 us at
// 1536 x 1024 x 8192 in 8 ranges of 16.
__device__ __forceinline__ bool split_of_block(int block, int splits, int tiles, int per, int min_per, int* split, int* tile) {
  const int groups = splits <= 8 ? 8 / splits : 0;  // XCDs per range of K
  if (per >= min_per && groups * splits == 8 && tiles % groups == 0) {
    const int x = block & 7, per_xcd = tiles / groups;
    *split = x % splits;
    *tile = (x / splits) * per_xcd + (block >> 3);
    return true;
  }
  *split = block / tiles;
  *tile = block % tiles;
  return false;
}

// A GEMM of 256 x 256 tiles split along K (gemm_w4_kernel with EPI = 4): a wave's 128 x 128 accumulators in the
// workspace, one 16-byte word per lane and 16 x 16 block (mb * 8 + nb), so that a wave's store or load of a block is one
// contiguous KiB.  gemm_w4_reduce_kernel: workgroup (tile, wave, half), wave v of it adds up block row mb = 4 half + v of
// that wave's accumulators over the ranges, always in the order of the ranges, and stores it as store_acc_rows does.
constexpr size_t kW4SlabFloats = 64 * 64 * 4;  // per wave
__device__ __forceinline__ f32x4* w4_slab(float* work, int split, int tiles, int tile, int wave) {
  return (f32x4*)(work + (((size_t)split * tiles + tile) * 4 + wave) * kW4SlabFloats);
}
template <bool SUM>  // SUM: the soak's instantiation, which adds the words it stores to g_gemm_reg_sum
__global__ __launch_bounds__(256) void gemm_w4_reduce_kernel(const float* __restrict__ work, int splits, int tiles, int tiles_n,
                                                             __bf16* __restrict__ C, size_t ldc) {
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, fr = lane & 15, fq = lane >> 4;
  const int tile = (int)blockIdx.x >> 3, wave = ((int)blockIdx.x >> 1) & 3, mb = ((int)blockIdx.x & 1) * 4 + (int)(threadIdx.x >> 6);
  const int tm = tile / tiles_n, tn = tile % tiles_n, wr = wave >> 1, wc = wave & 1;
  f32x4 acc[8];
#pragma unroll
  for (int nb = 0; nb < 8; nb++) acc[nb] = w4_slab((float*)work, 0, tiles, tile, wave)[(mb * 8 + nb) * 64 + lane];
  for (int s = 1; s < splits; s++) {
    const f32x4* slab = w4_slab((float*)work, s, tiles, tile, wave);
#pragma unroll
    for (int nb = 0; nb < 8; nb++) acc[nb] += slab[(mb * 8 + nb) * 64 + lane];
  }
  __bf16* const base = C + (size_t)(tm * BM + wr * 128 + fq * 32) * ldc + tn * BN + wc * 128 + fr * 8;
  unsigned sum = 0;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    u32x4 o;
#pragma unroll
    for (int h = 0; h < 4; h++) o[h] = __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{acc[2 * h][r], acc[2 * h + 1][r]}, bf16x2));
    __builtin_nontemporal_store(o, (u32x4*)(base + (size_t)(r * 8 + mb) * ldc));
    sum += (o[0] + o[1]) + (o[2] + o[3]);
  }
  if constexpr (SUM) reg_sum_add(sum, lane);
}

// What the LDS-DMA of a K-tile beyond the last one reads when the K-tile count is odd (the pipelines walk K-tiles in
// pairs): zeros for both operands, so that the pair's second half adds nothing.
__device__ uint4 g_gemm_zero_slab[264];  // 4 KiB + 128 B: the odd K-tile of a pair is read 128 bytes further on

template <int DT>
__global__ __launch_bounds__(kThreads, 2) void gemm_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                           __bf16* __restrict__ C, int M, int N, size_t row_bytes,
                                                           const uint32_t* __restrict__ SA, const uint32_t* __restrict__ SB, int) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;  // 2 x 4 waves

  // XCD-aware tile assignment (bijective for any grid size).
  const int nwg = gridDim.x, tiles_n = N / BN;
  const int xcd = blockIdx.x & 7, q = nwg >> 3, r = nwg & 7;
  const int wgid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (blockIdx.x >> 3);
  const int tm = wgid / tiles_n, tn = wgid % tiles_n;
  const char* gA = A + (size_t)tm * BM * row_bytes;
  const char* gB = B + (size_t)tn * BN * row_bytes;

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Per-lane fragment read offsets.  A fragment of 16 rows x 32 k-bytes... :
  //   lane reads row (lane&15), chunks (lane>>4) and 4 + (lane>>4): bf16 / fp8 one k-substep each, MX-fp8 the two
  //   halves of the single K=128 operand (registers 0-3: k 16*fq.., registers 4-7: k 64 + 16*fq..)
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
  {
    const int arow = wr * 128 + fr, brow = wc * 64 + fr;
    a_off[0] = lds_off<DT>(arow, fq);
    a_off[1] = lds_off<DT>(arow, 4 + fq);
    b_off[0] = lds_off<DT>(brow, fq);
    b_off[1] = lds_off<DT>(brow, 4 + fq);
  }

  const int nt = (int)(row_bytes / BKB);
  // MX: this wave's scale units are those of A row groups 4*tm + 2*wr + {0, 1} and B row group 4*tn + wc.
  const uint32_t* sa0 = nullptr;
  const uint32_t* sb0 = nullptr;
  if constexpr (DT == GMD_DT_MXFP8) {
    sa0 = SA + ((size_t)(4 * tm + 2 * wr) * nt) * kScaleUnitDwords + lane;
    sb0 = SB + ((size_t)(4 * tn + wc) * nt) * kScaleUnitDwords + lane;
  }
  stage<DT>(gA, row_bytes, 0, smem, wave, lane);
  stage<DT>(gB, row_bytes, 0, smem + kTileBytes, wave, lane);
  __syncthreads();  // with LDS-DMA in flight this also waits vmcnt(0)

  for (int t = 0; t < nt; t++) {
    char* cur = smem + (t & 1) * kBufBytes;
    char* nxt = smem + ((t + 1) & 1) * kBufBytes;
    if (t + 1 < nt) {  // prefetch the next K-step before touching this one
      stage<DT>(gA, row_bytes, (t + 1) * BKB, nxt, wave, lane);
      stage<DT>(gB, row_bytes, (t + 1) * BKB, nxt + kTileBytes, wave, lane);
    }
    const char* la = cur;
    const char* lb = cur + kTileBytes;
    if constexpr (DT != GMD_DT_MXFP8) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++) {
        i32x4 af[8], bfr[4];
#pragma unroll
        for (int nb = 0; nb < 4; nb++) bfr[nb] = *(const i32x4*)(lb + b_off[ks] + nb * 2048);
#pragma unroll
        for (int mb = 0; mb < 8; mb++) af[mb] = *(const i32x4*)(la + a_off[ks] + mb * 2048);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int mb = 0; mb < 8; mb++)
#pragma unroll
          for (int nb = 0; nb < 4; nb++) {
            if constexpr (DT == GMD_DT_BF16) {
              acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, bfr[nb]),
                                                                    __builtin_bit_cast(bf16x8, af[mb]), acc[mb][nb], 0, 0, 0);
            } else {
              acc[mb][nb] = mfma_fp8_half<0>(bfr[nb], af[mb], acc[mb][nb]);
              acc[mb][nb] = mfma_fp8_half<1>(bfr[nb], af[mb], acc[mb][nb]);
            }
          }
        __builtin_amdgcn_s_setprio(0);
      }
    } else {
      // MX-scaled fp8: one K=128 MFMA per tile pair; the scale byte of a 16-row block is picked by op_sel.
      const int sa_lo = (int)sa0[(size_t)t * kScaleUnitDwords], sa_hi = (int)sa0[((size_t)nt + t) * kScaleUnitDwords];
      const int sb = (int)sb0[(size_t)t * kScaleUnitDwords];
      i32x8 bfr[4];
#pragma unroll
      for (int nb = 0; nb < 4; nb++) {
        i32x4 lo = *(const i32x4*)(lb + b_off[0] + nb * 2048), hi = *(const i32x4*)(lb + b_off[1] + nb * 2048);
        bfr[nb] = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
      }
#pragma unroll
      for (int mb = 0; mb < 8; mb++) {
        i32x4 lo = *(const i32x4*)(la + a_off[0] + mb * 2048), hi = *(const i32x4*)(la + a_off[1] + mb * 2048);
        i32x8 af = i32x8{lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const int sa = mb < 4 ? sa_lo : sa_hi;
#pragma unroll
        for (int nb = 0; nb < 4; nb++) {
          // A goes first (the result's lane index then runs along N): the first scale is A's block mb & 3, the second B's block nb
          switch ((mb & 3) * 4 + nb) {
#define GMD_MX_CASE(MBQ, NBQ)                                                                                              \
  case MBQ * 4 + NBQ:                                                                                                      \
    acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(af, bfr[nb], acc[mb][nb], 0, 0, MBQ, sa, NBQ, sb);     \
    break;
            GMD_MX_CASE(0, 0) GMD_MX_CASE(0, 1) GMD_MX_CASE(0, 2) GMD_MX_CASE(0, 3)
            GMD_MX_CASE(1, 0) GMD_MX_CASE(1, 1) GMD_MX_CASE(1, 2) GMD_MX_CASE(1, 3)
            GMD_MX_CASE(2, 0) GMD_MX_CASE(2, 1) GMD_MX_CASE(2, 2) GMD_MX_CASE(2, 3)
            GMD_MX_CASE(3, 0) GMD_MX_CASE(3, 1) GMD_MX_CASE(3, 2) GMD_MX_CASE(3, 3)
#undef GMD_MX_CASE
          }
        }
      }
    }
    __syncthreads();  // next buffer landed (vmcnt(0)) and everyone is done reading this one
  }

  if constexpr (DT == GMD_DT_MXFP8) {
    // Rows dealt, A first: block (mb, nb), lane (fr, fq), register r is row 8 (4 fq + r) + mb of the wave's 128 and
    // column 8 fr + 4 (wc & 1) + nb of the 128 that waves wc and wc ^ 1 share: a lane's four nb are 8 contiguous bytes.
    __bf16* const base = C + (size_t)(tm * BM + wr * 128 + fq * 32) * N + tn * BN + 128 * (wc >> 1) + 8 * fr + 4 * (wc & 1);
#pragma unroll
    for (int mb = 0; mb < 8; mb++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const f32x4 v{acc[mb][0][r], acc[mb][1][r], acc[mb][2][r], acc[mb][3][r]};
        *(bf16x4*)(base + (size_t)(r * 8 + mb) * N) = __builtin_convertvector(v, bf16x4);
      }
  } else {
    // Epilogue: lane holds C[m][n..n+3] for m = .. + (lane&15), n = .. + (lane>>4)*4.
    store_acc<4, 1>(acc, C, (size_t)N, tm * BM + wr * 128 + fr, tn * BN + wc * 64, fq);
  }
}

// ---- (2b) the same GEMM as a phase-interleaved pipeline -------------------------
//
// Same tile, same LDS image and swizzle, same MFMA operand order as gemm_kernel;
// what changes is WHEN things happen.  The K loop runs in phases, four per K-tile,
// one C quadrant (64x32 of the wave's 128x64) each:
//
//     [ds_read the fragments this or the next phase needs] [issue one half-tile of
//     LDS-DMA] [s_waitcnt vmcnt(10)]  s_barrier  [16 MFMA (8 for MX-fp8)]  s_barrier
//
// * Operand tiles are staged in half-tiles of 128 rows x 128 B (16 KiB, two
//   LDS-DMA instructions per wave), in the order the phases consume them:
//   B-lo, A-lo, B-hi, A-hi of K-tile 0, B-lo of K-tile 1, ...  ("lo"/"hi" = the
//   rows that the first / second quadrant pair of every wave reads).  Half-tile j
//   is read in phase j-1 and lives in ring slot j % 8 (128 KiB in all).
// * Phase q issues half-tile q+7 and then waits until at most five half-tiles
//   (10 instructions) are outstanding, which retires half-tile q+2: everything a
//   phase reads was waited for one phase earlier, by every wave, before a barrier
//   the reader has passed.  The loop never drains to vmcnt(0).
// * Slot j % 8 is overwritten by half-tile j+8, issued in phase j+1: two phases
//   after the slot's only read, the margin the one-barrier stagger of the two
//   wave rows needs.  (Issue distances 5, 6 and 7 measure within 2 % of each
//   other at 8192^3; what bounds the kernel is shown by `make diag-ablate`:
//   the LDS-DMA stream alone and the MFMAs alone each take ~75 % of the full time.)
// * The two wave rows (wr = 0 / 1) share each SIMD and run one barrier apart, so
//   while one row issues its MFMAs the other row issues its LDS reads and DMA:
//   8 ds_read_b128 x 4 waves take the LDS pipe about as long as 16 MFMAs take the
//   matrix pipe.  The B-lo fragments of the next K-tile are read into whichever
//   B register set is idle, which makes the per-phase read counts 8 / 4 / 8 / 4.
constexpr int kHalfBytes = 128 * BKB;  // one ring slot
constexpr int kAheadDefault = 7;       // issue distance in half-tiles (5..7 with the 8-slot ring)

// Tile order: XCD-contiguous chunks (as above), walked in column-major groups of
// four tile rows, so the ~32 tiles an XCD has in flight form a 4 x 8 block that
// shares 4 A panels and 8 B panels in its L2 instead of 1 and 32.
__device__ __forceinline__ void tile_of(int wgid, int tiles_m, int tiles_n, int* tm, int* tn) {
#ifndef GMD_TILE_G
#define GMD_TILE_G 4  // tile rows per column-major group; other values are an experiment's business (-DGMD_TILE_G=)
#endif
  constexpr int G = GMD_TILE_G;
  const int per_group = G * tiles_n;
  const int group = wgid / per_group;
  const int first = group * G;
  const int gm = min(tiles_m - first, G);
  const int in = wgid - group * per_group;
  *tm = first + in % gm;
  *tn = in / gm;
}

// A fragment of the pipeline kernel is eight dwords: the two 16-byte LDS reads of one
// 16-row block (bf16: K sub-steps 0 and 1; MX-fp8: the two halves of the K=128 operand).
template <int MB0, int NB0, int DT>
__device__ __forceinline__ void quadrant(f32x4 (&acc)[8][4], const i32x8 (&a)[4], const i32x8 (&b)[2]) {
  if constexpr (DT == GMD_DT_BF16) {
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int mb = 0; mb < 4; mb++)
#pragma unroll
        for (int nb = 0; nb < 2; nb++) {
          const i32x4 bh = ks ? __builtin_shufflevector(b[nb], b[nb], 4, 5, 6, 7) : __builtin_shufflevector(b[nb], b[nb], 0, 1, 2, 3);
          const i32x4 ah = ks ? __builtin_shufflevector(a[mb], a[mb], 4, 5, 6, 7) : __builtin_shufflevector(a[mb], a[mb], 0, 1, 2, 3);
          acc[MB0 + mb][NB0 + nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
              __builtin_bit_cast(bf16x8, bh), __builtin_bit_cast(bf16x8, ah), acc[MB0 + mb][NB0 + nb], 0, 0, 0);
        }
  } else {
    static_assert(DT == GMD_DT_FP8, "the eight-wave pipelines run bf16 and unscaled fp8; MX-fp8 runs gemm_w4_kernel");
#pragma unroll
    for (int ks = 0; ks < 2; ks++)
#pragma unroll
      for (int h = 0; h < 2; h++)
#pragma unroll
        for (int mb = 0; mb < 4; mb++)
#pragma unroll
          for (int nb = 0; nb < 2; nb++) {
            const i32x4 bh = ks ? __builtin_shufflevector(b[nb], b[nb], 4, 5, 6, 7) : __builtin_shufflevector(b[nb], b[nb], 0, 1, 2, 3);
            const i32x4 ah = ks ? __builtin_shufflevector(a[mb], a[mb], 4, 5, 6, 7) : __builtin_shufflevector(a[mb], a[mb], 0, 1, 2, 3);
            acc[MB0 + mb][NB0 + nb] = h ? mfma_fp8_half<1>(bh, ah, acc[MB0 + mb][NB0 + nb]) : mfma_fp8_half<0>(bh, ah, acc[MB0 + mb][NB0 + nb]);
          }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

#define GMD_PHASE_FENCE()                  \
  __builtin_amdgcn_sched_barrier(0);       \
  __builtin_amdgcn_s_barrier();            \
  __builtin_amdgcn_sched_barrier(0)

#ifdef GMD_GEMM_ABLATE
// Clock stamps of the ablation builds: per workgroup, shader-clock ticks and 100 MHz
// real-time ticks spent in the kernel (ABL bit 4).  Nothing else reads this buffer.
constexpr int kStampSlots = 4096;
__device__ unsigned long long g_gemm_stamps[2 * kStampSlots];
// Persistent grid with stamps: per tile, 100 MHz real-time ticks at the top of the tile loop, after the wait
// for the first half-tiles (and the previous tile's stores), after the last MFMA, and the shader ticks of the K loop.
__device__ unsigned long long g_gemm_timeline[4 * kStampSlots];
#endif

// ABL is for timing-only ablation builds (make diag-ablate; results are wrong by design):
// bit 0 drops the fragment reads of the loop, bit 1 the LDS-DMA and its waits, bit 2 the MFMAs;
// bit 3 points every workgroup at the operands of tile (0, 0) (an always-L2-resident stream).
// EPI: see store_acc.
template <int DT, int ABL = 0, int AHEAD = kAheadDefault, int EPI = 1>
__global__ __launch_bounds__(kThreads) void gemm_pp_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                           __bf16* __restrict__ C, int M, int N, size_t row_bytes,
                                                           const uint32_t*, const uint32_t*, int) {  // no scales: bf16 / fp8 only
  constexpr int kAhead = AHEAD;  // half-tiles issued ahead of the running phase; kAhead - 2 of them are in flight
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // Tiles are dealt to the XCDs in contiguous chunks (bijective for any count), one workgroup per tile.
  const int tiles_m = M / BM, tiles_n = N / BN;
  const int total = (int)gridDim.x;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8n = total & 7;
  const int wgid = (xcd < r8n ? xcd * (q8 + 1) : r8n * (q8 + 1) + (xcd - r8n) * q8) + (int)(blockIdx.x >> 3);
  int tm, tn;
  tile_of(wgid, tiles_m, tiles_n, &tm, &tn);
#ifdef GMD_GEMM_ABLATE
  unsigned long long stamp_t0 = 0, stamp_r0 = 0;
  if constexpr (ABL & 16) {
    stamp_t0 = __builtin_amdgcn_s_memtime();
    stamp_r0 = __builtin_amdgcn_s_memrealtime();
  }
#endif
  // ABL bit 3: every workgroup streams the operands of tile (0, 0), so all but the first fetch of a line hit L2.
  const char* gA = A + (size_t)((ABL & 8) ? 0 : tm) * BM * row_bytes;
  const char* gB = B + (size_t)((ABL & 8) ? 0 : tn) * BN * row_bytes;

  // Staging: this wave fills pieces 2*wave and 2*wave+1 (8 rows x 128 B each) of every slot.
  // Slot-local row l maps to tile row (l>>6)*128 + (l&63) [+64 for hi] in A and (l>>5)*64 + (l&31) [+32 for hi] in B.
  const int r8 = lane >> 3;
  const uint32_t src_chunk = (uint32_t)(((lane & 7) ^ chunk_swizzle<DT>(r8)) * 16);
  uint32_t off_a[2], off_b[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int l = (2 * wave + i) * 8 + r8;
    off_a[i] = (uint32_t)(((l >> 6) * 128 + (l & 63)) * row_bytes) + src_chunk;
    off_b[i] = (uint32_t)(((l >> 5) * 64 + (l & 31)) * row_bytes) + src_chunk;
  }
  const size_t hi_a = 64 * row_bytes, hi_b = 32 * row_bytes;
  char* const my_pieces = smem + wave * 2048;

  // kind: 0 B-lo, 1 A-lo, 2 B-hi, 3 A-hi;  par: K-tile parity;  kt: K-tile index
  auto issue = [&](auto KIND, auto PAR, int kt) {
    constexpr int kind = decltype(KIND)::value, par = decltype(PAR)::value;
    const char* base = ((kind & 1) ? gA + (kind == 3 ? hi_a : 0) : gB + (kind == 2 ? hi_b : 0)) + (size_t)kt * BKB;
    // Keep the wave-uniform part of the address whole, in scalar registers: the DMA then takes "SGPR base +
    // 32-bit lane offset" and the loop carries 4 offset VGPRs instead of eight 64-bit lane pointers.
    asm volatile("" : "+s"(base));
    char* dst = my_pieces + (par * 4 + kind) * kHalfBytes;
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((kind & 1) ? off_a[i] : off_b[i])),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };

  // Fragment read offsets inside a slot (local rows: A wr*64 + fr, B wc*32 + fr).
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int ch = ks * 4 + fq;
    a_off[ks] = lds_off<DT>(wr * 64 + fr, ch);
    b_off[ks] = lds_off<DT>(wc * 32 + fr, ch);
  }
  auto read_a = [&](i32x8 (&a)[4], int slot) {
#pragma unroll
    for (int mb = 0; mb < 4; mb++) {
      const i32x4 lo = *(const i32x4*)(smem + slot * kHalfBytes + a_off[0] + mb * 2048);
      const i32x4 hi = *(const i32x4*)(smem + slot * kHalfBytes + a_off[1] + mb * 2048);
      a[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto read_b = [&](i32x8 (&b)[2], int slot) {
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const i32x4 lo = *(const i32x4*)(smem + slot * kHalfBytes + b_off[0] + nb * 2048);
      const i32x4 hi = *(const i32x4*)(smem + slot * kHalfBytes + b_off[1] + nb * 2048);
      b[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  f32x4 acc[8][4];
  i32x8 a[4], bp[2], bq[2];
  if constexpr (ABL & 1) {  // the loop never loads them: give them defined contents
#pragma unroll
    for (int i = 0; i < 4; i++) a[i] = i32x8{lane, 1, 2, 3, 4, 5, 6, 7};
    bq[0] = bq[1] = i32x8{7, 6, 5, 4, 3, 2, 1, lane};
  }

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  // Prologue of a tile: half-tiles 0..kAhead-1 in flight (all of K-tile 0 and the start of K-tile 1).
  static_assert(kAhead >= 5 && kAhead <= 7, "ring of 8 slots: 2 phases of read-to-overwrite margin need kAhead <= 7");
  auto issue_prologue = [&]() {
    issue(I0{}, I0{}, 0);
    issue(I1{}, I0{}, 0);
    issue(I2{}, I0{}, 0);
    issue(I3{}, I0{}, 0);
    issue(I0{}, I1{}, 1);
    if constexpr (kAhead > 5) issue(I1{}, I1{}, 1);
    if constexpr (kAhead > 6) issue(I2{}, I1{}, 1);
  };

  // One phase.  P: phase within a pair of K-tiles (0..7).  TAIL: last pair, where issue stops and the waits shrink.
  auto phase = [&](auto PH, auto TAILF, int kt0) {
    constexpr int P = decltype(PH)::value;
    constexpr bool TAIL = decltype(TAILF)::value != 0;
    constexpr int par = P >> 2, p = P & 3;  // parity of the K-tile this phase computes on
    // -- reads (half-tile P+1 of the pair) --
    if constexpr (!(ABL & 1)) {
      if constexpr (p == 0) read_a(a, par * 4 + 1);
      else if constexpr (p == 1) read_b(par ? bp : bq, par * 4 + 2);
      else if constexpr (p == 2) read_a(a, par * 4 + 3);
      else if constexpr (!(TAIL && P == 7)) read_b(par ? bp : bq, (par ^ 1) * 4 + 0);
    }
    // -- issue half-tile P+5 of the pair, i.e. kind (P+5)&3 of K-tile kt0 + (P+5)/4 --
    constexpr int J = P + kAhead;
    if constexpr (ABL & 2) {
    } else if constexpr (!(TAIL && J >= 8)) {
      issue(std::integral_constant<int, (J & 3)>{}, std::integral_constant<int, ((J >> 2) & 1)>{}, kt0 + (J >> 2));
      wait_vmcnt<2 * (kAhead - 2)>();
    } else if constexpr (P <= 5) {
      // Last pair, nothing left to issue: half-tile P+2 must land, 7-(P+2) half-tiles may still fly.
      wait_vmcnt<2 * (5 - P)>();
    }
    GMD_PHASE_FENCE();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    // -- MFMAs of quadrant p: (m-lo,n-lo) (m-lo,n-hi) (m-hi,n-hi) (m-hi,n-lo); B-lo sits in bp on even K-tiles, bq on odd --
    if constexpr (ABL & 4) {  // keep what the MFMAs would have consumed alive
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("" ::"v"(a[i]));
#pragma unroll
      for (int i = 0; i < 2; i++) asm volatile("" ::"v"(bp[i]), "v"(bq[i]));
    } else if constexpr (p == 0) quadrant<0, 0, DT>(acc, a, par ? bq : bp);
    else if constexpr (p == 1) quadrant<0, 2, DT>(acc, a, par ? bp : bq);
    else if constexpr (p == 2) quadrant<4, 2, DT>(acc, a, par ? bp : bq);
    else quadrant<4, 0, DT>(acc, a, par ? bq : bp);
    __builtin_amdgcn_s_setprio(0);
    GMD_PHASE_FENCE();
  };

  const int npairs = (int)(row_bytes / BKB) >> 1;
  using F = std::integral_constant<int, 0>;
  using T = std::integral_constant<int, 1>;

  issue_prologue();
#pragma unroll
  for (int i = 0; i < 8; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  wait_vmcnt<2 * (kAhead - 2)>();  // half-tiles 0 and 1 landed
  GMD_PHASE_FENCE();
  read_b(bp, 0);  // B-lo of K-tile 0
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (wr == 1) { GMD_PHASE_FENCE(); }  // run the second wave row one barrier behind the first

  for (int it = 0; it + 1 < npairs; it++) {
    const int kt0 = 2 * it;
    phase(std::integral_constant<int, 0>{}, F{}, kt0);
    phase(std::integral_constant<int, 1>{}, F{}, kt0);
    phase(std::integral_constant<int, 2>{}, F{}, kt0);
    phase(std::integral_constant<int, 3>{}, F{}, kt0);
    phase(std::integral_constant<int, 4>{}, F{}, kt0);
    phase(std::integral_constant<int, 5>{}, F{}, kt0);
    phase(std::integral_constant<int, 6>{}, F{}, kt0);
    phase(std::integral_constant<int, 7>{}, F{}, kt0);
  }
  {
    const int kt0 = 2 * (npairs - 1);
    phase(std::integral_constant<int, 0>{}, T{}, kt0);
    phase(std::integral_constant<int, 1>{}, T{}, kt0);
    phase(std::integral_constant<int, 2>{}, T{}, kt0);
    phase(std::integral_constant<int, 3>{}, T{}, kt0);
    phase(std::integral_constant<int, 4>{}, T{}, kt0);
    phase(std::integral_constant<int, 5>{}, T{}, kt0);
    phase(std::integral_constant<int, 6>{}, T{}, kt0);
    phase(std::integral_constant<int, 7>{}, T{}, kt0);
  }
  if (wr == 0) { GMD_PHASE_FENCE(); }  // the barrier the second wave row took up front
#ifdef GMD_GEMM_ABLATE
  if constexpr (ABL & 16) {
    asm volatile("s_nop 0" ::"v"(acc[7][3]));  // stamp behind the last MFMA's result
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && wgid < kStampSlots) {
      g_gemm_stamps[2 * wgid] = t1 - stamp_t0;
      g_gemm_stamps[2 * wgid + 1] = r1 - stamp_r0;
    }
  }
#endif
  store_acc<4, EPI>(acc, C, (size_t)N, tm * BM + wr * 128 + fr, tn * BN + wc * 64, fq);
}

// ---- (2b') the eight-wave pipeline as one unbroken stream over a workgroup's tiles ----
//
// One workgroup per CU walks its tiles as in the persistent form of gemm_pp_kernel, but the ring never
// drains between tiles: half-tile numbering simply continues, so the last K-tile pair of a tile already
// issues the first seven half-tiles of the next tile into the slots they would have had anyway, and its last
// phase reads the next tile's first B fragments.  Between the last MFMA of one tile and the first of the next
// a wave only converts and stores its accumulators; it waits for neither the stores nor any fetch.
//
// The stores share the wave's vector-memory counter with the LDS-DMA, and the counter retires in issue
// order.  Behind an epilogue the sequence a wave has outstanding is
//     [half-tiles q+3 .. 6] [kStores stores] [half-tiles 7 .. q+7]        (q = phases since the epilogue)
// so for q = 0..4 "all but the youngest 10 + kStores" is exactly "half-tiles up to q+2 have landed", with the
// stores free to drain behind the MFMAs; from q = 5 on the wait is the steady-state one (by then the stores are
// ~1.6 us old).  The two wave rows keep their one-barrier stagger across tiles: a row stores while the other
// row runs its MFMAs.
template <int DT, int EPI = 1, int ABL = 0>  // ABL 16: clock stamps and per-tile timeline (ablation builds only)
__global__ __launch_bounds__(kThreads) void gemm_ps_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                           __bf16* __restrict__ C, int M, int N, size_t row_bytes,
                                                           const uint32_t*, const uint32_t*, int) {  // no scales: bf16 / fp8 only
  constexpr int kAhead = kAheadDefault;
  static_assert(kAhead == 7, "the mode tables below are written for an issue distance of seven half-tiles");
  constexpr int kStores = EPI ? 16 : 32;  // store instructions per wave of one tile (store_acc)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 2, wc = wave & 3;

  // XCD x owns a contiguous chunk of tiles; its workgroup j walks tiles j, j + G/8, ... of the chunk.
  const int tiles_m = M / BM, tiles_n = N / BN, total = tiles_m * tiles_n;
  const int xcd = blockIdx.x & 7, q8 = total >> 3, r8n = total & 7;
  const int chunk0 = xcd < r8n ? xcd * (q8 + 1) : r8n * (q8 + 1) + (xcd - r8n) * q8;
  const int chunk_end = chunk0 + q8 + (xcd < r8n ? 1 : 0);
  const int tile_step = max((int)gridDim.x >> 3, 1);
  int wgid = chunk0 + (int)(blockIdx.x >> 3);
  if (wgid >= chunk_end) return;  // whole workgroup, before any barrier
  int tm, tn;
  tile_of(wgid, tiles_m, tiles_n, &tm, &tn);
  const char* gA = A + (size_t)tm * BM * row_bytes;
  const char* gB = B + (size_t)tn * BN * row_bytes;
  const char* nA = gA;
  const char* nB = gB;

  // Staging and fragment addressing: as in gemm_pp_kernel.
  const int r8 = lane >> 3;
  const uint32_t src_chunk = (uint32_t)(((lane & 7) ^ chunk_swizzle<DT>(r8)) * 16);
  uint32_t off_a[2], off_b[2];
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int l = (2 * wave + i) * 8 + r8;
    off_a[i] = (uint32_t)(((l >> 6) * 128 + (l & 63)) * row_bytes) + src_chunk;
    off_b[i] = (uint32_t)(((l >> 5) * 64 + (l & 31)) * row_bytes) + src_chunk;
  }
  const size_t hi_a = 64 * row_bytes, hi_b = 32 * row_bytes;
  char* const my_pieces = smem + wave * 2048;
  // kind: 0 B-lo, 1 A-lo, 2 B-hi, 3 A-hi;  par: ring half;  kt: K-tile index inside the tile whose panels pa / pb are
  auto issue = [&](auto KIND, auto PAR, const char* pa, const char* pb, int kt) {
    constexpr int kind = decltype(KIND)::value, par = decltype(PAR)::value;
    const char* base = ((kind & 1) ? pa + (kind == 3 ? hi_a : 0) : pb + (kind == 2 ? hi_b : 0)) + (size_t)kt * BKB;
    asm volatile("" : "+s"(base));  // wave-uniform part of the address in scalar registers
    char* dst = my_pieces + (par * 4 + kind) * kHalfBytes;
#pragma unroll
    for (int i = 0; i < 2; i++)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + ((kind & 1) ? off_a[i] : off_b[i])),
                                       (__attribute__((address_space(3))) void*)(dst + i * 1024), 16, 0, 0);
  };
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int ch = ks * 4 + fq;
    a_off[ks] = lds_off<DT>(wr * 64 + fr, ch);
    b_off[ks] = lds_off<DT>(wc * 32 + fr, ch);
  }
  auto read_a = [&](i32x8 (&a)[4], int slot) {
#pragma unroll
    for (int mb = 0; mb < 4; mb++) {
      const i32x4 lo = *(const i32x4*)(smem + slot * kHalfBytes + a_off[0] + mb * 2048);
      const i32x4 hi = *(const i32x4*)(smem + slot * kHalfBytes + a_off[1] + mb * 2048);
      a[mb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };
  auto read_b = [&](i32x8 (&b)[2], int slot) {
#pragma unroll
    for (int nb = 0; nb < 2; nb++) {
      const i32x4 lo = *(const i32x4*)(smem + slot * kHalfBytes + b_off[0] + nb * 2048);
      const i32x4 hi = *(const i32x4*)(smem + slot * kHalfBytes + b_off[1] + nb * 2048);
      b[nb] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    }
  };

  f32x4 acc[8][4];
  i32x8 a[4], bp[2], bq[2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  // One phase of a K-tile pair.  P: phase (0..7).  MODE: 0 inside a tile; 1 last pair of the workgroup's last tile
  // (issue stops, waits shrink); 2 last pair of a tile with a successor (half-tiles 8.. are the successor's 0..);
  // 3 first pair behind an epilogue (the stores may stay outstanding for five phases).
  auto phase = [&](auto PH, auto MODEC, int kt0) {
    constexpr int P = decltype(PH)::value, MODE = decltype(MODEC)::value;
    constexpr int par = P >> 2, p = P & 3;
    if constexpr (p == 0) read_a(a, par * 4 + 1);
    else if constexpr (p == 1) read_b(par ? bp : bq, par * 4 + 2);
    else if constexpr (p == 2) read_a(a, par * 4 + 3);
    else if constexpr (!(MODE == 1 && P == 7)) read_b(par ? bp : bq, (par ^ 1) * 4 + 0);
    constexpr int J = P + kAhead;
    using KIND = std::integral_constant<int, (J & 3)>;
    using PAR = std::integral_constant<int, ((J >> 2) & 1)>;
    if constexpr (J < 8 || MODE == 0 || MODE == 3) {
      issue(KIND{}, PAR{}, gA, gB, kt0 + (J >> 2));
      wait_vmcnt<2 * (kAhead - 2) + ((MODE == 3 && P <= 4) ? kStores : 0)>();
    } else if constexpr (MODE == 2) {
      issue(KIND{}, PAR{}, nA, nB, (J >> 2) - 2);
      wait_vmcnt<2 * (kAhead - 2)>();
    } else if constexpr (P <= 5) {
      wait_vmcnt<2 * (5 - P)>();  // nothing left to issue: half-tile P+2 must land, 7-(P+2) may still fly
    }
    GMD_PHASE_FENCE();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(1);
    if constexpr (p == 0) quadrant<0, 0, DT>(acc, a, par ? bq : bp);
    else if constexpr (p == 1) quadrant<0, 2, DT>(acc, a, par ? bp : bq);
    else if constexpr (p == 2) quadrant<4, 2, DT>(acc, a, par ? bp : bq);
    else quadrant<4, 0, DT>(acc, a, par ? bq : bp);
    __builtin_amdgcn_s_setprio(0);
    GMD_PHASE_FENCE();
  };
  auto pair = [&](auto MODEC, int kt0) {
    phase(std::integral_constant<int, 0>{}, MODEC, kt0);
    phase(std::integral_constant<int, 1>{}, MODEC, kt0);
    phase(std::integral_constant<int, 2>{}, MODEC, kt0);
    phase(std::integral_constant<int, 3>{}, MODEC, kt0);
    phase(std::integral_constant<int, 4>{}, MODEC, kt0);
    phase(std::integral_constant<int, 5>{}, MODEC, kt0);
    phase(std::integral_constant<int, 6>{}, MODEC, kt0);
    phase(std::integral_constant<int, 7>{}, MODEC, kt0);
  };
  using M0 = std::integral_constant<int, 0>;
  using M1 = std::integral_constant<int, 1>;
  using M2 = std::integral_constant<int, 2>;
  using M3 = std::integral_constant<int, 3>;
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>;

  const int npairs = (int)(row_bytes / BKB) >> 1;  // >= 2 (checked by the launcher)
  // Prologue of the first tile: half-tiles 0..6 in flight, 0 and 1 landed, B-lo of K-tile 0 in registers.
  issue(I0{}, I0{}, gA, gB, 0);
  issue(I1{}, I0{}, gA, gB, 0);
  issue(I2{}, I0{}, gA, gB, 0);
  issue(I3{}, I0{}, gA, gB, 0);
  issue(I0{}, I1{}, gA, gB, 1);
  issue(I1{}, I1{}, gA, gB, 1);
  issue(I2{}, I1{}, gA, gB, 1);
  zero_acc();
  wait_vmcnt<2 * (kAhead - 2)>();
  GMD_PHASE_FENCE();
  read_b(bp, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (wr == 1) { GMD_PHASE_FENCE(); }  // run the second wave row one barrier behind the first

#ifdef GMD_GEMM_ABLATE
  unsigned long long tl_top = 0, tl_t0 = 0;
  if constexpr (ABL & 16) {
    tl_top = __builtin_amdgcn_s_memrealtime();
    tl_t0 = __builtin_amdgcn_s_memtime();
  }
#endif
  pair(M0{}, 0);  // first pair of the first tile; every later tile's first pair runs at the bottom of the loop
  for (;;) {
    for (int it = 1; it + 1 < npairs; it++) pair(M0{}, 2 * it);
    const int next = wgid + tile_step;
    if (next >= chunk_end) break;
    int ntm, ntn;
    tile_of(next, tiles_m, tiles_n, &ntm, &ntn);
    nA = A + (size_t)ntm * BM * row_bytes;
    nB = B + (size_t)ntn * BN * row_bytes;
    pair(M2{}, 2 * (npairs - 1));
#ifdef GMD_GEMM_ABLATE
    if constexpr (ABL & 16) {
      asm volatile("s_nop 0" ::"v"(acc[7][3]));  // stamp behind the last MFMA's result
      const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
      if (threadIdx.x == 0 && wgid < kStampSlots) {
        g_gemm_timeline[4 * wgid] = tl_top;
        g_gemm_timeline[4 * wgid + 1] = tl_top;
        g_gemm_timeline[4 * wgid + 2] = r1;
        g_gemm_timeline[4 * wgid + 3] = t1 - tl_t0;
      }
    }
#endif
    store_acc<4, EPI>(acc, C, (size_t)N, tm * BM + wr * 128 + fr, tn * BN + wc * 64, fq);
    __builtin_amdgcn_sched_barrier(0);
    zero_acc();
    tm = ntm;
    tn = ntn;
    gA = nA;
    gB = nB;
    wgid = next;
#ifdef GMD_GEMM_ABLATE
    if constexpr (ABL & 16) {
      tl_top = __builtin_amdgcn_s_memrealtime();
      tl_t0 = __builtin_amdgcn_s_memtime();
    }
#endif
    pair(M3{}, 0);
  }
  pair(M1{}, 2 * (npairs - 1));
  if (wr == 0) { GMD_PHASE_FENCE(); }  // the barrier the second wave row took up front
#ifdef GMD_GEMM_ABLATE
  if constexpr (ABL & 16) {
    asm volatile("s_nop 0" ::"v"(acc[7][3]));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && wgid < kStampSlots) {
      g_gemm_timeline[4 * wgid] = tl_top;
      g_gemm_timeline[4 * wgid + 1] = tl_top;
      g_gemm_timeline[4 * wgid + 2] = r1;
      g_gemm_timeline[4 * wgid + 3] = t1 - tl_t0;
    }
  }
#endif
  store_acc<4, EPI>(acc, C, (size_t)N, tm * BM + wr * 128 + fr, tn * BN + wc * 64, fq);
}

// ---- (2c) four waves, 128 x 128 of C each ---------------------------------------
//
// In cycles the eight-wave pipeline above is within 2 % of its MFMAs alone; what it
// loses is clock: the chip is at its power cap and holds ~1.73 GHz under it against
// ~2.38 GHz for the MFMAs alone, and the LDS fragment reads are the largest draw
// (scripts/gpu_gemm_clock.py).  A wave that owns 128 x 128 of C instead of 128 x 64
// reads (128 + 128) rows per K-tile for twice the MFMAs: a third fewer LDS bytes per
// FLOP.  That takes 256 accumulator registers, so one wave per SIMD, four per
// workgroup, and no partner wave to hide behind: every phase interleaves its own
// memory operations between its own MFMAs.
//
// * Same LDS image, swizzle and 8-slot ring of 16 KiB half-tiles as above, in the
//   order A-lo, B-lo, B-hi, A-hi per K-tile ("lo"/"hi": the first / second 64 rows
//   of each wave's 128).  Four phases per K-tile, one 64 x 64 quadrant each:
//   (lo,lo) (lo,hi) (hi,hi) (hi,lo), 32 MFMAs (16 for MX-fp8) = 512 matrix cycles.
// * Phase g computes on fragments already in registers, and between its first
//   twelve MFMAs issues the 8 ds_read_b128 of half-tile g+2 (into the fragment set
//   that is idle: two A sets, two B sets that swap roles with K-tile parity) and the
//   4 LDS-DMA instructions of half-tile g+8.  Every second phase ends with
//   vmcnt(16) (half-tiles up to g+4 landed, four still in flight), lgkmcnt(0) and
//   a barrier; the phases in between end with nothing.
// * Slot (g+8) % 8 was last read in phase g-2, at least one barrier earlier.
// * With one wave per SIMD there is nothing to issue from while this wave issues an
//   instruction that is not an MFMA: every scalar or vector ALU instruction, s_nop and
//   avoidable s_waitcnt between two MFMAs is matrix-pipe time (3 to 6 cycles each,
//   512 phases per 8192-deep tile).  The K loop therefore holds MFMAs, fragment reads,
//   DMA pieces, one M0 write per half-tile and one wait per phase, and nothing else:
//   LDS read bases for the far half of the ring in registers of their own, half-tile
//   bases computed once per pair of K-tiles (the odd K-tile goes through the
//   instruction's immediate), M0 written three MFMAs ahead of the piece that needs
//   it, the LDS wait stated once per phase in a form the compiler can see.
//   284 k -> 272 k cycles per tile on bf16 (262 k of MFMA issue).
// * Measured (8192^3 bf16, one rocprofv3 pass): 701.0 us and 1,219,217 cycles
//   against hipBLASLt's 704.6 us and 1,221,069; 1-2 % ahead of the eight-wave
//   kernels on every data type.
constexpr int kThreadsW4 = 256;
constexpr int kAheadW4 = 8;

// The 64 accumulators fill the 256 AGPRs exactly and must be updated in place.  Written
// with the builtin, hipcc (ROCm 7.2) spreads them over VGPRs and AGPRs and moves them
// about with ~2 v_accvgpr copies per MFMA; the "+a" constraint pins each one to its
// AGPR quad.  An MFMA whose C operand is the previous MFMA's whole result needs no wait
// states; the epilogue's reads come after W4_DRAIN.
// `sub`: bf16 k-substep (0 / 1); fp8 k-substep * 2 + half (0..3); unused for MX-fp8, whose scale dwords
// sa / sb hold four E8M0 bytes each and whose op_sel fields pick byte MBQ (A, second operand) and NBQ (B, first).
// UNIT (with DT MX-fp8): no scales; the same K=128 instruction in its unscaled form.
template <int DT, int MBQ, int NBQ, bool UNIT = false>
__device__ __forceinline__ void mfma_step_t(f32x4& acc, const i32x8& a, const i32x8& b, int sub, int sa, int sb) {
  if constexpr (DT == GMD_DT_BF16) {
    const i32x4 bh = sub ? __builtin_shufflevector(b, b, 4, 5, 6, 7) : __builtin_shufflevector(b, b, 0, 1, 2, 3);
    const i32x4 ah = sub ? __builtin_shufflevector(a, a, 4, 5, 6, 7) : __builtin_shufflevector(a, a, 0, 1, 2, 3);
    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(bh), "v"(ah));
  } else if constexpr (DT == GMD_DT_FP8) {
    typedef int i32x2 __attribute__((ext_vector_type(2)));
    const i32x2 bh = sub == 0 ? __builtin_shufflevector(b, b, 0, 1) : sub == 1 ? __builtin_shufflevector(b, b, 2, 3)
                     : sub == 2 ? __builtin_shufflevector(b, b, 4, 5) : __builtin_shufflevector(b, b, 6, 7);
    const i32x2 ah = sub == 0 ? __builtin_shufflevector(a, a, 0, 1) : sub == 1 ? __builtin_shufflevector(a, a, 2, 3)
                     : sub == 2 ? __builtin_shufflevector(a, a, 4, 5) : __builtin_shufflevector(a, a, 6, 7);
    asm volatile("v_mfma_f32_16x16x32_fp8_fp8 %0, %1, %2, %0" : "+a"(acc) : "v"(bh), "v"(ah));
  } else if constexpr (UNIT) {
    asm volatile("v_mfma_f32_16x16x128_f8f6f4 %0, %1, %2, %0" : "+a"(acc) : "v"(b), "v"(a));
  } else {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel:[%5,%6,0] op_sel_hi:[%7,%8,0]"
                 : "+a"(acc)
                 : "v"(b), "v"(a), "v"(sb), "v"(sa), "n"(NBQ & 1), "n"(MBQ & 1), "n"(NBQ >> 1), "n"(MBQ >> 1));
  }
}
template <int DT, bool UNIT = false>
__device__ __forceinline__ void mfma_step(f32x4& acc, const i32x8& a, const i32x8& b, int sub, int mbq, int nbq, int sa, int sb) {
  if constexpr (DT != GMD_DT_MXFP8 || UNIT) {
    mfma_step_t<DT, 0, 0, UNIT>(acc, a, b, sub, 0, 0);
  } else {
    switch (mbq * 4 + nbq) {  // constant once the caller's loop is unrolled
#define GMD_W4_CASE(MBQ, NBQ) \
  case MBQ * 4 + NBQ: mfma_step_t<DT, MBQ, NBQ>(acc, a, b, 0, sa, sb); break;
      GMD_W4_CASE(0, 0) GMD_W4_CASE(0, 1) GMD_W4_CASE(0, 2) GMD_W4_CASE(0, 3)
      GMD_W4_CASE(1, 0) GMD_W4_CASE(1, 1) GMD_W4_CASE(1, 2) GMD_W4_CASE(1, 3)
      GMD_W4_CASE(2, 0) GMD_W4_CASE(2, 1) GMD_W4_CASE(2, 2) GMD_W4_CASE(2, 3)
      GMD_W4_CASE(3, 0) GMD_W4_CASE(3, 1) GMD_W4_CASE(3, 2) GMD_W4_CASE(3, 3)
#undef GMD_W4_CASE
    }
  }
}
#define W4_DRAIN() asm volatile("s_nop 15\n\ts_nop 15" ::: "memory")
// s_waitcnt lgkmcnt(0) as the builtin (vmcnt and expcnt fields at their maxima: not waited for): unlike the same
// instruction in inline asm the compiler's own wait-count pass takes it into account.
__device__ __forceinline__ void w4_wait_lds() { __builtin_amdgcn_s_waitcnt(0xC07F); }

// ABL (ablation builds only; results are wrong, timing is the point): 1 no fragment reads, 2 no DMA, 4 no MFMAs,
// 8 no M0 writes (what the one per half-tile costs),
// 16 clock stamps, 32 no barriers, 64 (with 16) the stamps split the prologue, 128 (with 16) the stamps say when and on which CU instead of how long.  EPI: see store_acc.
// STREAM: one workgroup per CU walks the tiles j, j + G/8, ... of its XCD's chunk without draining the ring: the
// half-tiles a tile's last pair issues and reads beyond its end are the first ones of the next tile (they land in
// the slots and registers that tile's first pair expects), so a tile boundary is the epilogue and nothing else.
// Measured 1-3 % slower than one workgroup per tile: the one vmcnt counter retires loads, stores and LDS-DMA in order,
// so the first half-tile issued behind a tile's stores waits for the store burst of all CUs to drain, which the ring
// covers only in part, while a workgroup that ends after its stores never waits for them.  Kept as an A/B reference.
// GEN: any M, any N that is a multiple of 8 (bf16, fp8), any number of K-tiles.  Rows beyond the matrix are fetched
// from its last row (their products land in accumulators that are never stored), stores are bounds-tested, and an odd
// K-tile count is completed by a K-tile of zeros.  The full-tile, even-count instantiation pays for none of it.
// UNIT (DT MX-fp8 only): unscaled fp8 on the K=128 instruction.  The LDS image, the fragment layout and the interleaving
// are MX-fp8's; there are no scale tensors, no scale ring and no fifth memory instruction per half-tile.  This is how
// GMD_DT_FP8 runs: the K=128 instruction does in 32 cycles what four of the K=32 fp8 instruction do in 64.
template <int DT, int ABL = 0, int EPI = 1, bool STREAM = false, bool GEN = false, bool UNIT = false>
__global__ __launch_bounds__(kThreadsW4) void gemm_w4_kernel(const char* __restrict__ A, const char* __restrict__ B,
                                                             __bf16* __restrict__ C, int M, int N, size_t row_bytes,
                                                             const uint32_t* __restrict__ SA, const uint32_t* __restrict__ SB, int ldc) {
  // ldc: C's row stride in elements; N where the launch covers the whole matrix, larger where it covers the leading
  // columns of a wider one (launch_gemm_t).  Only the epilogue reads it.
  constexpr bool kMx = DT == GMD_DT_MXFP8;    // the K=128 instruction and its layouts
  constexpr bool kScaled = kMx && !UNIT;      // ... with scale tensors
  static_assert(!UNIT || kMx, "UNIT is a form of the MX-fp8 kernel");
  constexpr int kPerHalf = kScaled ? 5 : 4;  // vector-memory instructions a wave issues per half-tile (MX: + its scale unit)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;
#ifdef GMD_GEMM_ABLATE
  unsigned long long stamp_t0 = 0, stamp_r0 = 0;
  if constexpr (ABL & 16) {
    stamp_t0 = __builtin_amdgcn_s_memtime();
    stamp_r0 = __builtin_amdgcn_s_memrealtime();
  }
#endif

  // Per-tile launch: workgroup = tile.  Stream: XCD x owns a contiguous chunk of tiles, its workgroup j walks tiles
  // j, j + G/8, ... of the chunk.
  static_assert(!(GEN && STREAM), "the stream walks full tiles and even K-tile counts only");
  // EPI = 4: the GEMM is split along K.  The grid is `tiles` workgroups per range of K-tiles; workgroup b walks `ldc`
  // K-tiles (the last range what is left) of tile b % tiles from K-tile (b / tiles) * ldc on and leaves its fp32
  // accumulators, as they are, in the workspace that C points to (gemm_w4_reduce_kernel adds the ranges up and stores C).
  constexpr bool kSplit = EPI == 4;
  static_assert(!kSplit || (!STREAM && !GEN && ABL == 0), "the split runs whole tiles and pairs of K-tiles");
  const int tiles_m = GEN ? (M + BM - 1) / BM : M / BM, tiles_n = GEN ? (N + BN - 1) / BN : N / BN;
  const int nunits = STREAM || kSplit ? tiles_m * tiles_n : (int)gridDim.x;
  int bid = (int)blockIdx.x, split = 0;
  bool split_by_xcd = false;
  if constexpr (kSplit) split_by_xcd = split_of_block((int)blockIdx.x, (int)gridDim.x / nunits, nunits, ldc, 8, &split, &bid);
  const int kt_first = kSplit ? split * ldc : 0;
  const int kcount = kSplit ? min(ldc, (int)(row_bytes / BKB) - kt_first) : (int)(row_bytes / BKB);  // K-tiles this workgroup walks
  const int xcd = bid & 7, q8 = nunits >> 3, r8n = nunits & 7;
  const int chunk0 = xcd < r8n ? xcd * (q8 + 1) : r8n * (q8 + 1) + (xcd - r8n) * q8;
  const int chunk_end = chunk0 + q8 + (xcd < r8n ? 1 : 0);
  const int tile_step = STREAM ? max((int)gridDim.x >> 3, 1) : 1;
  int wgid = split_by_xcd ? bid : chunk0 + (bid >> 3);
  if (STREAM && wgid >= chunk_end) return;  // whole workgroup, before any barrier
  int tm, tn;
  tile_of(wgid, tiles_m, tiles_n, &tm, &tn);
  // Tile bases carry the offset bias of the DMA lanes (see off_imm) from the start, so that no phase has to subtract it.
  constexpr int kOffBias = 3 * 1024;
  const char* gA = A + (size_t)tm * BM * row_bytes - kOffBias + (size_t)kt_first * BKB;
  const char* gB = B + (size_t)tn * BN * row_bytes - kOffBias + (size_t)kt_first * BKB;
  // the tile after this one (stream only): where the last pair's surplus half-tiles come from
  bool has_next = false;
  int ntm = tm, ntn = tn;
  const char* nA = gA;
  const char* nB = gB;
  auto look_ahead = [&]() {
    has_next = STREAM && wgid + tile_step < chunk_end;
    if (has_next) {
      tile_of(wgid + tile_step, tiles_m, tiles_n, &ntm, &ntn);
      nA = A + (size_t)ntm * BM * row_bytes - kOffBias;
      nB = B + (size_t)ntn * BN * row_bytes - kOffBias;
    }
  };
  look_ahead();
  // Staging: this wave fills pieces 4*wave .. 4*wave+3 (8 rows x 128 B each) of every slot, for A and B alike.
  // The narrow epilogue (EPI 0, an A/B reference for bf16 and fp8): slot-local row l is tile row (l>>6)*128 + (l&63)
  // [+64 for hi].  Otherwise (kRowsDealt): slot-local row l = 64 w + 16 c + f is tile row 128 w + 8 f + c [+4 for hi], so that
  // fragment lane f of block c (c + 4 in the hi half) holds row or column 8 f + c of the wave's 128: with A as the
  // MFMA's first operand the result lane (fr, fq), register r of block (mb, nb) is row (4 fq + r) * 8 + mb, column
  // fr * 8 + nb, and the epilogue stores whole rows from neighbouring lanes (store_acc_rows).  The fragment reads and
  // the LDS image are the same either way; only the global row a DMA lane fetches differs.
  constexpr bool kRowsDealt = EPI != 0;
  const int r8 = lane >> 3;
  const uint32_t src_chunk = (uint32_t)(((lane & 7) ^ chunk_swizzle<DT>(r8)) * 16);
  constexpr int kHiRows = kRowsDealt ? 4 : 64;
  uint32_t off[4], off_b[4];  // A and B fetch the same rows of their tiles, unless one of them has rows to clamp (GEN)
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int l = (4 * wave + i) * 8 + r8;
    const int tile_row = kRowsDealt ? (l >> 6) * 128 + (l & 15) * 8 + ((l & 63) >> 4) : (l >> 6) * 128 + (l & 63);
    off[i] = (uint32_t)(tile_row * row_bytes) + src_chunk;
    off_b[i] = off[i];
    if constexpr (GEN && kRowsDealt) {
      // M and N are multiples of 8 and the rows are dealt in groups of eight (8 f + c: c < 4 in the lo half-tile, the same
      // lane offset on a base four rows further on in the hi one), so a group is inside the matrix or outside as a whole:
      // a lane whose group is outside fetches the last group inside instead.
      const int rows_a = M - tm * BM, rows_b = N - tn * BN;
      const int row_a = (tile_row & ~7) >= rows_a ? rows_a - 8 + (tile_row & 7) : tile_row;
      const int row_b = (tile_row & ~7) >= rows_b ? rows_b - 8 + (tile_row & 7) : tile_row;
      off[i] = (uint32_t)(row_a * row_bytes) + src_chunk;
      off_b[i] = (uint32_t)(row_b * row_bytes) + src_chunk;
    }
  }
  const size_t hi_off = (size_t)kHiRows * row_bytes;
  char* const my_pieces = smem + wave * 4096;

  // kind: 0 A-lo, 1 B-lo, 2 B-hi, 3 A-hi.  Wave-uniform part of the source address of one half-tile, kept in SGPRs.
  auto piece_base = [&](int kind, int kt, const char* pa, const char* pb) {
    const char* base = ((kind == 0 || kind == 3) ? pa : pb) + ((kind >= 2) ? hi_off : 0) + (size_t)kt * BKB;
    asm volatile("" : "+s"(base));
    return base;
  };
  // One LDS-DMA instruction per piece and nothing else: the source is SGPR base + 32-bit VGPR offset + immediate, the
  // LDS side M0 + the SAME immediate, so the four pieces of a half-tile share one M0 write (piece i's VGPR offset has
  // i KiB taken off to make up for the immediate on the global side).  Through the builtin every piece costs a 64-bit
  // VALU add and an M0 write of its own, and with one wave per SIMD a VALU instruction is an MFMA issue slot lost
  // (measured: 12 k of 293 k cycles per tile).
  // The VGPR offset is unsigned: a piece whose first row is low in the tile (any piece but the first, once the rows are
  // dealt) would wrap below zero, so every offset carries a bias that the SGPR base gives back (piece_base).
  uint32_t off_imm[4], off_imm_b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    off_imm[i] = off[i] + (uint32_t)(kOffBias - i * 1024);
    off_imm_b[i] = off_b[i] + (uint32_t)(kOffBias - i * 1024);
  }
  // GEN: the zero K-tile is read with one offset for all four pieces (the immediate moves on by 1 KiB per piece; 4 KiB in all).
  const uint32_t off_zero = (uint32_t)(kOffBias + lane * 16);
  const char* const zero_base = (const char*)g_gemm_zero_slab - kOffBias;
  // `zero`: GEN only, wave-uniform: this half-tile is the zero K-tile.  `is_b`: a compile-time constant at every call.
  // PAR: the K-tile's parity.  The two K-tiles of a pair share their four bases (one per kind): the odd one is the same
  // base read 128 bytes further on, through the immediate, with 128 taken off M0 to leave the LDS side where it was.
  // One scalar 64-bit add per kind and PAIR instead of one per half-tile: with one wave per SIMD scalar instructions
  // between the MFMAs are not free either (seven add pairs fewer per pair of K-tiles were 3.8 k cycles per tile).
  auto dma = [my_pieces, o0 = off_imm[0], o1 = off_imm[1], o2 = off_imm[2], o3 = off_imm[3], b0 = off_imm_b[0], b1 = off_imm_b[1],
              b2 = off_imm_b[2], b3 = off_imm_b[3], off_zero](const char* base, int slot, auto I, auto PAR, bool is_b, bool zero,
                                               auto EARLY_M0) {
    constexpr int i = decltype(I)::value;
    constexpr int kOdd = decltype(PAR)::value * BKB;
    const uint32_t lds = (uint32_t)(uintptr_t)(my_pieces + slot * kHalfBytes) - kOdd;
    uint32_t o = i == 0 ? o0 : i == 1 ? o1 : i == 2 ? o2 : o3;
    if constexpr (GEN) {
      if (is_b) o = i == 0 ? b0 : i == 1 ? b1 : i == 2 ? b2 : b3;
      if (zero) o = off_zero;
    }
    // Only piece 0 names M0 as clobbered.  Pieces 1..3 rely on the M0 it wrote and say nothing about it: named there too,
    // the clobber makes the hazard recognizer put a wait state between two pieces that are only two MFMAs apart (24 s_nop
    // per pair of K-tiles), and nothing the compiler generates in this loop writes M0 (scripts/check_gemm_codegen.py counts
    // the M0 writes between the first and the last MFMA of every instantiation).
    if constexpr (i == 0 && !(ABL & 8) && !decltype(EARLY_M0)::value)
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(o), "s"(base), "n"(kOdd), "s"(lds)
                   : "memory", "m0");
    else
      asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(o), "s"(base), "n"(i * 1024 + kOdd) : "memory");
  };
  // MX scales: the packed unit (256 B: 64 rows x 4 blocks of one K-tile) that belongs to a half-tile travels with
  // it, by one 4-byte-per-lane LDS-DMA into this wave's own slot of a small ring behind the operand ring.
  const int ktiles = (int)(row_bytes / BKB);
  char* const my_scales = smem + kLdsBytes + wave * 8 * 256;
  auto scale_base = [&](const uint32_t* S, int t, int w) { return S + (size_t)(4 * t + 2 * w) * ktiles * kScaleUnitDwords; };
  // (a split GEMM's range of K begins kt_first units into the tile's run of units)
  const uint32_t* sc_a = kScaled ? scale_base(SA, tm, wr) + (size_t)kt_first * kScaleUnitDwords : nullptr;  // wave-uniform
  const uint32_t* sc_b = kScaled ? scale_base(SB, tn, wc) + (size_t)kt_first * kScaleUnitDwords : nullptr;
  const uint32_t* nsc_a = kScaled && has_next ? scale_base(SA, ntm, wr) : sc_a;
  const uint32_t* nsc_b = kScaled && has_next ? scale_base(SB, ntn, wc) : sc_b;
  uint32_t lane_x4 = (uint32_t)lane * 4;
  asm volatile("" : "+v"(lane_x4));
  // Source of one scale unit, worked out once per pair and kind (fetch_pair) and held in SGPRs: the distance from the
  // tensor's start goes through readfirstlane (a no-op where the compiler has followed the uniformity, one VALU instruction
  // where it has not) and back onto the kernel argument.
  auto scale_src = [&](int kind, int kt, const uint32_t* sa_base, const uint32_t* sb_base) {
    const bool is_a = kind == 0 || kind == 3;
    const uint32_t* src = (is_a ? sa_base : sb_base) + ((size_t)(kind >= 2 ? ktiles : 0) + kt) * kScaleUnitDwords;
    const uint32_t dist = __builtin_amdgcn_readfirstlane((uint32_t)((const char*)src - (const char*)(is_a ? SA : SB)));
    src = (const uint32_t*)((const char*)(is_a ? SA : SB) + dist);
    asm volatile("" : "+s"(src));
    return src;
  };
  // One unit: SGPR base + one VGPR offset (4 * lane) that never changes (a per-lane 64-bit address would be a VALU add per
  // unit).  PAR: as for the operands, the odd K-tile's unit is the even one's base read 256 bytes further on, through the
  // immediate, with 256 taken off M0.  GEN with an odd count: the zero K-tile takes the last K-tile's scales (`same`).
  auto dma_scale = [my_scales, lane_x4](const uint32_t* src, int slot, auto PAR, bool same) {
    constexpr int kOdd = decltype(PAR)::value * 256;
    uint32_t lds = (uint32_t)(uintptr_t)(my_scales + slot * 256) - kOdd;
    if constexpr (GEN && kOdd != 0) {
      if (same) {  // never taken outside the general instantiation's last real pair
        asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1" ::"v"(lane_x4), "s"(src), "s"(lds + kOdd) : "memory", "m0");
        return;
      }
    }
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 offset:%3" ::"v"(lane_x4), "s"(src), "s"(lds), "n"(kOdd)
                 : "memory", "m0");
  };
  // The same in two statements for the K loop, M0 three MFMAs ahead of the load (see set_m0); not for GEN, whose zero
  // K-tile changes the unit's place.
  auto set_scale_m0 = [my_scales](int slot, auto PAR) {
    const uint32_t lds = (uint32_t)(uintptr_t)(my_scales + slot * 256) - decltype(PAR)::value * 256;
    asm volatile("s_mov_b32 m0, %0" ::"s"(lds) : "memory", "m0");
  };
  auto scale_load = [lane_x4](const uint32_t* src, auto PAR) {
    asm volatile("global_load_lds_dword %0, %1 offset:%2" ::"v"(lane_x4), "s"(src), "n"(decltype(PAR)::value * 256) : "memory");
  };
  auto read_scale = [&](int slot) { return *(const int*)(my_scales + slot * 256 + lane * 4); };

  // Fragment read offsets inside a slot (local rows: A wr*64 + fr, B wc*64 + fr).
  const int fr = lane & 15, fq = lane >> 4;
  // ds_read's immediate offset is 16 bits and the ring is 128 KiB: slots 4..7 get bases of their own, 64 KiB further
  // on, that an empty asm makes opaque.  Left to itself the compiler keeps one base and adds 65536 to it in front of the
  // reads of half the phases: eight VALU adds per pair of K-tiles, each of them an MFMA issue slot with one wave per SIMD.
  int a_off[2], b_off[2], a_off_far[2], b_off_far[2];
#pragma unroll
  for (int ks = 0; ks < 2; ks++) {
    const int ch = ks * 4 + fq;
    a_off[ks] = lds_off<DT>(wr * 64 + fr, ch);
    b_off[ks] = lds_off<DT>(wc * 64 + fr, ch);
    a_off_far[ks] = a_off[ks] + 4 * kHalfBytes;
    b_off_far[ks] = b_off[ks] + 4 * kHalfBytes;
    asm volatile("" : "+v"(a_off_far[ks]), "+v"(b_off_far[ks]));
  }
  // One of the eight 16-byte reads of a half-tile's fragments: block j>>1 (16 rows), half j&1.
  auto read_one = [&](i32x8 (&f)[4], int slot, bool is_a, int j) {
    const int base = slot >= 4 ? (is_a ? a_off_far[j & 1] : b_off_far[j & 1]) : (is_a ? a_off[j & 1] : b_off[j & 1]);
    const i32x4 v = *(const i32x4*)(smem + (slot & 3) * kHalfBytes + base + (j >> 1) * 2048);
    if (j & 1) f[j >> 1].hi = v;
    else f[j >> 1].lo = v;
  };

  // MX-fp8 clears the accumulators in the shadow of the prologue's DMA issue (eight of them behind every half-tile, held
  // there by an empty asm): issue is bound by the address path at ~60 cycles per instruction and per wave, and the 256
  // v_accvgpr_write otherwise sit between the wait for the first data and the first MFMA (1 k cycles of a 148 k-cycle
  // tile).  For bf16 and fp8 the compiler's own placement costs nothing measurable and the interleaving slows the issue.
  f32x4 acc[8][8];
  if constexpr (!kMx) {
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
  i32x8 aL[4], aH[4], bX[4], bY[4];
  int sAL = 0, sAH = 0, sBX = 0, sBY = 0;  // MX: the scale dword that goes with each fragment set

  // Prologue: half-tiles 0..7 (K-tiles 0 and 1) in flight, 0..3 landed (phases 0 and 1 read 2 and 3), 0 and 1 in registers.
#ifdef GMD_GEMM_ABLATE
  unsigned long long stamp_pa = 0, stamp_pb = 0, stamp_pc = 0;  // ABL 64: before the first DMA, behind the last, data landed
  if constexpr ((ABL & 80) == 80) {
    asm volatile("s_nop 0" ::"v"(off_imm[0]), "v"(a_off[0]), "s"(gA), "s"(gB));
    stamp_pa = __builtin_amdgcn_s_memtime();
  }
#endif
  // In the K loop M0 is written three MFMAs ahead of piece 0 (EARLY_M0): the write needs a wait state before the load that
  // uses it, and three MFMAs are one, where an s_nop of its own between two MFMAs is an issue slot.
  auto set_m0 = [my_pieces](int slot, auto PAR) {
    const uint32_t lds = (uint32_t)(uintptr_t)(my_pieces + slot * kHalfBytes) - decltype(PAR)::value * BKB;
    asm volatile("s_mov_b32 m0, %0" ::"s"(lds) : "memory", "m0");
  };
  // What the DMA of one pair of K-tiles needs, worked out once per pair (fetch_pair): the four bases at the pair's even
  // K-tile, whether its odd K-tile is the zero one (GEN, odd count), and the MX scale tensors and K-tile index.
  const int last_kt = kcount - 1;
  const char* pair_base[4];
  bool pair_zero_odd = false;
  const uint32_t* pair_sa = sc_a;
  const uint32_t* pair_sb = sc_b;
  const uint32_t* pair_scale[4] = {nullptr, nullptr, nullptr, nullptr};  // MX: the scale unit of each kind at the pair's even K-tile
  auto fetch_pair = [&](int bk) {  // bk: the pair's even K-tile
    const char* pa = gA;
    const char* pb = gB;
    pair_sa = sc_a, pair_sb = sc_b;
    if (bk > last_kt) {  // beyond the tile: the next tile's first pair (stream), or the last pair again, into slots nobody reads
      if (has_next) {
        bk -= last_kt + 1;
        pa = nA, pb = nB, pair_sa = nsc_a, pair_sb = nsc_b;
      } else {
        bk = GEN ? (last_kt & ~1) : last_kt - 1;
      }
    }
    pair_zero_odd = GEN && bk + 1 > last_kt;
#pragma unroll
    for (int kind = 0; kind < 4; kind++) {
      pair_base[kind] = piece_base(kind, bk, pa, pb);
      if constexpr (kScaled) pair_scale[kind] = scale_src(kind, bk, pair_sa, pair_sb);
    }
  };
  // One half-tile's DMA outside the K loop's interleaving (the prologue).
  auto issue_half = [&](auto J) {
    constexpr int j = decltype(J)::value, kind = j & 3;
    using Par = std::integral_constant<int, (j >> 2)>;
    const bool zero = GEN && Par::value == 1 && pair_zero_odd;
    const char* base = pair_base[kind];
    if constexpr (GEN) base = zero ? zero_base : base;
    dma(base, j, std::integral_constant<int, 0>{}, Par{}, kind == 1 || kind == 2, zero, std::false_type{});
    dma(base, j, std::integral_constant<int, 1>{}, Par{}, kind == 1 || kind == 2, zero, std::false_type{});
    dma(base, j, std::integral_constant<int, 2>{}, Par{}, kind == 1 || kind == 2, zero, std::false_type{});
    dma(base, j, std::integral_constant<int, 3>{}, Par{}, kind == 1 || kind == 2, zero, std::false_type{});
    if constexpr (kScaled) dma_scale(pair_scale[kind], j, Par{}, zero);
    if constexpr (kMx) {
#pragma unroll
      for (int t = 0; t < 8; t++) {
        acc[j][t] = f32x4{0.f, 0.f, 0.f, 0.f};
        asm volatile("" : "+a"(acc[j][t]));
      }
    }
  };
  fetch_pair(0);
  issue_half(std::integral_constant<int, 0>{});
  issue_half(std::integral_constant<int, 1>{});
  issue_half(std::integral_constant<int, 2>{});
  issue_half(std::integral_constant<int, 3>{});
  issue_half(std::integral_constant<int, 4>{});
  issue_half(std::integral_constant<int, 5>{});
  issue_half(std::integral_constant<int, 6>{});
  issue_half(std::integral_constant<int, 7>{});
#ifdef GMD_GEMM_ABLATE
  if constexpr ((ABL & 80) == 80) stamp_pb = __builtin_amdgcn_s_memtime();
#endif
  wait_vmcnt<kPerHalf * (kAheadW4 - 4)>();
#ifdef GMD_GEMM_ABLATE
  if constexpr ((ABL & 80) == 80) stamp_pc = __builtin_amdgcn_s_memtime();
#endif
  GMD_PHASE_FENCE();
#pragma unroll
  for (int j = 0; j < 8; j++) read_one(aL, 0, true, j);
#pragma unroll
  for (int j = 0; j < 8; j++) read_one(bX, 1, false, j);
  if constexpr (kScaled) {
    sAL = read_scale(0);
    sBX = read_scale(1);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  GMD_PHASE_FENCE();
#ifdef GMD_GEMM_ABLATE
  unsigned long long stamp_s1 = 0, stamp_s2 = 0;  // shader ticks: prologue done, K loop done (before the last wait)
  if constexpr (ABL & 16) stamp_s1 = __builtin_amdgcn_s_memtime();
#endif

  // One phase.  P: phase within a pair of K-tiles (0..7).  There is ONE loop body and no specialised last
  // iteration: hipcc allocates a separate tail separately and bridges the two with v_accvgpr copies placed
  // right against the asm MFMAs, whose wait states it cannot see (measured: wrong sums).  The last pair
  // therefore keeps issuing (the last K-tile again, into slots nobody reads any more; those fetches hit in L2 and have
  // landed by the end of the pair: the wait behind the loop measures 136 cycles) and keeps reading (fragments nobody
  // multiplies), which also keeps every vmcnt immediate the same.  Branching round the surplus DMA under a wave-uniform
  // flag was tried: four s_cmp + s_cbranch per phase, never taken before the last pair, cost 10 % of the K loop.
  // So was pointing the first half of the surplus at the first K-tile of the tile that follows on this XCD (ids + 32), to
  // have it in L2 for that workgroup's prologue: the prologue stayed at 3.2-3.4 k cycles and the K loop grew by 2.4 k.
  // And so was running the surplus DMA with no lane enabled (EXEC = 0: the instruction keeps its place in the vmcnt queue,
  // scripts/probes/exec0_vmcnt_probe.hip, and moves no data): the two EXEC writes per piece cost the same 10 %.
  constexpr int kStoresW4 = EPI ? 32 : 64;  // store instructions per wave of one tile (store_acc<8>)
  static_assert(!STREAM || kPerHalf * (kAheadW4 - 4) + kStoresW4 <= 63, "vmcnt is a six-bit counter");
  auto phase = [&](auto PH, int kt0, bool after_epilogue) {
    constexpr int P = decltype(PH)::value;
    constexpr int par = P >> 2, p = P & 3;
    constexpr int MB0 = (p >= 2) ? 4 : 0, NB0 = (p == 1 || p == 2) ? 4 : 0;
    i32x8(&Af)[4] = (p < 2) ? aL : aH;
    i32x8(&Bf)[4] = (p == 0 || p == 3) ? (par ? bY : bX) : (par ? bX : bY);  // B-lo sits in bX on even K-tiles, bY on odd
    // read during this phase: half-tile P+2 of the pair
    constexpr int R = P + 2, rk = R & 3, rpar = (R >> 2) & 1;
    i32x8(&Rf)[4] = rk == 0 ? aL : rk == 3 ? aH : (rk == 1) == (rpar == 0) ? bX : bY;
    constexpr int rslot = R & 7;
    // issued during this phase: half-tile P+8, into the slot half-tile P leaves, from the bases fetch_pair(kt0 + 2) set
    // (beyond the end of the tile: the next tile's first pair (stream, with a successor) or the last pair again).
    // GEN, odd K-tile count: the K-tile right behind the last one is the zero K-tile, for A and B alike.
    const bool zero = GEN && par == 1 && pair_zero_odd;
    const char* base = pair_base[p];
    if constexpr (GEN) base = zero ? zero_base : base;
    constexpr bool is_b = p == 1 || p == 2;
    using Par = std::integral_constant<int, par>;
    int& sAf = (p < 2) ? sAL : sAH;
    int& sBf = (p == 0 || p == 3) ? (par ? sBY : sBX) : (par ? sBX : sBY);
    int& sRf = rk == 0 ? sAL : rk == 3 ? sAH : (rk == 1) == (rpar == 0) ? sBX : sBY;
    constexpr int NM = DT == GMD_DT_BF16 ? 32 : DT == GMD_DT_FP8 ? 64 : 16;
    // The fragments this phase multiplies were read at least 24 MFMAs ago.  One wait the compiler can see says so at the
    // top of the phase (an even phase starts behind the odd one's wait before the barrier); without it the compiler
    // counts the reads down with four or five s_waitcnt of its own between the MFMAs of every phase.
    if constexpr (P & 1) w4_wait_lds();
#pragma unroll
    for (int i = 0; i < NM; i++) {
      const int sub = i >> 4, mb = (i >> 2) & 3, nb = i & 3;
      if constexpr (!(ABL & 4)) {
        // mfma_step's first matrix operand is its `b`: rows dealt -> A goes first (result lanes run along N)
        if constexpr (kRowsDealt) mfma_step<DT, UNIT>(acc[MB0 + mb][NB0 + nb], Bf[nb], Af[mb], sub, nb, mb, sBf, sAf);
        else mfma_step<DT, UNIT>(acc[MB0 + mb][NB0 + nb], Af[mb], Bf[nb], sub, mb, nb, sAf, sBf);
      }
      if constexpr (kMx) {
        // sixteen MFMAs per phase: reads behind the first eight, then the four pieces, the scale unit, the scale read
        // M0 for the pieces goes out behind MFMA 5, M0 for the scale unit behind piece 3, each three MFMAs ahead of its load.
        if (i < 8) {
          if constexpr (!(ABL & 1)) read_one(Rf, rslot, rk == 0 || rk == 3, i);
          if constexpr (!(ABL & 2)) {
            if (i == 5) set_m0(P, Par{});
          }
        } else if (i < 12) {
          if constexpr (!(ABL & 2)) {
            if (i == 8) dma(base, P, std::integral_constant<int, 0>{}, Par{}, is_b, zero, std::true_type{});
            else if (i == 9) dma(base, P, std::integral_constant<int, 1>{}, Par{}, is_b, zero, std::true_type{});
            else if (i == 10) dma(base, P, std::integral_constant<int, 2>{}, Par{}, is_b, zero, std::true_type{});
            else dma(base, P, std::integral_constant<int, 3>{}, Par{}, is_b, zero, std::true_type{});
            if constexpr (kScaled && !GEN) {
              if (i == 11) set_scale_m0(P, Par{});
            }
          }
        } else if (i == 13) {
          if constexpr (kScaled && !(ABL & 1)) sRf = read_scale(rslot);
        } else if (i == 14) {
          if constexpr (kScaled && !(ABL & 2)) {
            if constexpr (GEN) dma_scale(pair_scale[p], P, Par{}, zero);
            else scale_load(pair_scale[p], Par{});
          }
        }
        if (i < (kScaled ? 15 : 12)) __builtin_amdgcn_sched_barrier(0);
      } else {
        // 32 (bf16) or 64 (fp8) MFMAs per phase: fragment read j behind MFMA j*kReadStep, DMA piece q behind MFMA
        // kDma0 + q*kDmaStep.  With one wave per SIMD a memory instruction right behind another one costs MFMA issue
        // slots; pieces two MFMAs apart measured 1.6 % fewer cycles per tile than back to back.
        constexpr int SCHED = ABL >> 8;
        constexpr int kReadStep = (SCHED == 2 || SCHED == 4) ? 2 : 1;
        constexpr int kDma0 = SCHED == 0 ? 8 : SCHED == 1 ? 8 : SCHED == 2 ? 17 : SCHED == 3 ? 12 : 16;
        constexpr int kDmaStep = SCHED == 0 ? 2 : SCHED == 1 ? 1 : SCHED == 2 ? 4 : SCHED == 3 ? 5 : 3;
        static_assert(7 * kReadStep < NM && kDma0 + 3 * kDmaStep < NM, "schedule does not fit the phase");
        if (i % kReadStep == 0 && i / kReadStep < 8) {
          if constexpr (!(ABL & 1)) read_one(Rf, rslot, rk == 0 || rk == 3, i / kReadStep);
        }
        static_assert(kDma0 >= 3, "M0 is written three MFMAs ahead of the first piece");
        if (i == kDma0 - 3) {
          if constexpr (!(ABL & 2) && !(ABL & 8)) set_m0(P, Par{});
        }
        if (i >= kDma0 && (i - kDma0) % kDmaStep == 0 && (i - kDma0) / kDmaStep < 4) {
          if constexpr (!(ABL & 2)) {
            const int q = (i - kDma0) / kDmaStep;
            if (q == 0) dma(base, P, std::integral_constant<int, 0>{}, Par{}, is_b, zero, std::true_type{});
            else if (q == 1) dma(base, P, std::integral_constant<int, 1>{}, Par{}, is_b, zero, std::true_type{});
            else if (q == 2) dma(base, P, std::integral_constant<int, 2>{}, Par{}, is_b, zero, std::true_type{});
            else dma(base, P, std::integral_constant<int, 3>{}, Par{}, is_b, zero, std::true_type{});
          }
        }
        if (i <= (kDma0 + 3 * kDmaStep > 7 * kReadStep ? kDma0 + 3 * kDmaStep : 7 * kReadStep)) __builtin_amdgcn_sched_barrier(0);
      }
    }
    if constexpr (kScaled) asm volatile("" ::"v"(sAf), "v"(sBf));
    // Keep both operand sets whole until the phase's last MFMA, so that no half of them is handed to one of
    // this phase's ds_reads or address computations while MFMAs that read it are still to issue.
#pragma unroll
    for (int j = 0; j < 4; j++) asm volatile("" ::"v"(Af[j]), "v"(Bf[j]));
    // With one wave per SIMD every barrier is exposed matrix-pipe time (~45 cycles), so there is one per TWO
    // phases: after odd phase h, half-tiles up to h+4 have landed (h+1 and h+2 read h+3 and h+4) and the reads
    // of phases h-1 and h are done, which frees the slots that phases h+1 and h+2 refill.
    if constexpr (P & 1) {
      // Behind an epilogue the tile's stores sit between half-tiles 7 and 8 in the memory queue: the waits of phases
      // 1 and 3 are for half-tiles older than the stores and let them stay outstanding; from phase 5 on the awaited
      // half-tile is younger than the stores and the plain count is the right one.
      if (STREAM && (P == 1 || P == 3) && after_epilogue) wait_vmcnt<kPerHalf * (kAheadW4 - 4) + kStoresW4>();
      else wait_vmcnt<kPerHalf * (kAheadW4 - 4)>();
      w4_wait_lds();
      if constexpr (!(ABL & 32)) GMD_PHASE_FENCE();
    } else {
      __builtin_amdgcn_sched_barrier(0);
    }
  };

  const int npairs = (kcount + (GEN ? 1 : 0)) >> 1;
  // The launcher sends no shape with fewer than one pair.  Saying so removes the path round the K loop, on which the
  // accumulators are constants, and with it the copies that merge the two paths at the top of the block behind the loop:
  // those read accumulator registers ahead of the drain.
  __builtin_assume(npairs > 0);
  bool after_epilogue = false;
  for (;;) {
    for (int it = 0; it < npairs; it++) {
      const int kt0 = 2 * it;
      const bool ae = after_epilogue && it == 0;
      fetch_pair(kt0 + 2);
      phase(std::integral_constant<int, 0>{}, kt0, ae);
      phase(std::integral_constant<int, 1>{}, kt0, ae);
      phase(std::integral_constant<int, 2>{}, kt0, ae);
      phase(std::integral_constant<int, 3>{}, kt0, ae);
      phase(std::integral_constant<int, 4>{}, kt0, ae);
      phase(std::integral_constant<int, 5>{}, kt0, ae);
      phase(std::integral_constant<int, 6>{}, kt0, ae);
      phase(std::integral_constant<int, 7>{}, kt0, ae);
    }
    if constexpr (!STREAM) break;
    // Tile boundary: the next tile's half-tiles 0..7 are in flight or landed, its fragments 0 and 1 in registers.
    // The last tile goes through the same code: a second copy of the epilogue behind the loop would have its
    // conversions merged with this one's and hoisted above the branch between them, all 128 results live at once.
    __builtin_amdgcn_sched_barrier(0);
    W4_DRAIN();
    __builtin_amdgcn_sched_barrier(0);
    // The accumulators are made opaque first: with the K loop's "never entered" path in view the compiler folds
    // the conversions into the two incoming edges (constants on one, all 128 results ahead of the first store on
    // the other) and spills the K loop's own registers to make room.
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) asm volatile("" : "+a"(acc[i][j]));
    if constexpr (kRowsDealt) store_acc_rows<EPI, true>(acc, C, (size_t)ldc, tm * BM + wr * 128, tn * BN + wc * 128, fr, fq);
    else store_acc<8, EPI, true>(acc, C, (size_t)ldc, tm * BM + wr * 128 + fr, tn * BN + wc * 128, fq);
    if (!has_next) break;
#pragma unroll
    for (int i = 0; i < 8; i++)
#pragma unroll
      for (int j = 0; j < 8; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    W4_DRAIN();
    __builtin_amdgcn_sched_barrier(0);
    wgid += tile_step;
    tm = ntm, tn = ntn, gA = nA, gB = nB;
    if constexpr (kScaled) sc_a = nsc_a, sc_b = nsc_b;
    look_ahead();
    if constexpr (kScaled) {
      nsc_a = has_next ? scale_base(SA, ntm, wr) : sc_a;
      nsc_b = has_next ? scale_base(SB, ntn, wc) : sc_b;
    }
    after_epilogue = true;
  }
#ifdef GMD_GEMM_ABLATE
  if constexpr (ABL & 16) {
    asm volatile("s_nop 0" ::"v"(a_off[0]));
    stamp_s2 = __builtin_amdgcn_s_memtime();
  }
#endif
  // One statement: the wait for the last pair's surplus DMA (it must not outlive the workgroup's LDS) and the drain of
  // the matrix pipe, whose result latency the compiler does not know.  The copies out of the accumulators that the
  // epilogue needs are the compiler's to place; it places them behind the first statement of this block, which is why
  // the drain is part of that statement (tests/test_diag_codegen.py reads the disassembly and holds it to that).
  asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
#ifdef GMD_GEMM_ABLATE
  unsigned long long stamp_s3 = 0;
  if constexpr (ABL & 16) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamp_s3 = t1;
    if (threadIdx.x == 0 && wgid < kStampSlots) {
      g_gemm_stamps[2 * wgid] = t1 - stamp_t0;
      g_gemm_stamps[2 * wgid + 1] = r1 - stamp_r0;
    }
  }
#endif

  if constexpr (!STREAM) {
    // The lane's position is worked out afresh (two mbcnt): carried over from the top of the kernel it is two registers
    // that live through the K loop for nothing, and that the register allocator may decide to spill and reload here.
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int fr_e = lane_e & 15, fq_e = lane_e >> 4;
    if constexpr (kSplit) {
      f32x4* const slab = w4_slab((float*)C, split, nunits, tm * tiles_n + tn, wave);
#pragma unroll
      for (int mb = 0; mb < 8; mb++)
#pragma unroll
        for (int nb = 0; nb < 8; nb++) slab[(mb * 8 + nb) * 64 + lane_e] = acc[mb][nb];
    } else if constexpr (kRowsDealt && GEN) store_acc_rows_edge<EPI>(acc, C, (size_t)ldc, tm * BM + wr * 128, tn * BN + wc * 128, fr_e, fq_e, M, N);
    else if constexpr (kRowsDealt) store_acc_rows<EPI>(acc, C, (size_t)ldc, tm * BM + wr * 128, tn * BN + wc * 128, fr_e, fq_e);
    else store_acc<8, EPI>(acc, C, (size_t)ldc, tm * BM + wr * 128 + fr, tn * BN + wc * 128, fq);
  }
#ifdef GMD_GEMM_ABLATE
  if constexpr (ABL & 16) {
    // Timeline of the tile in shader ticks from the kernel's first instruction: prologue done, K loop done, last
    // DMA landed, last store issued.
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long s4 = __builtin_amdgcn_s_memtime();
    if constexpr ((ABL & 192) == 192) {
      // per wave: ticks from the first instruction to the end of the prologue's wait for data (how far apart the four waves are)
      if (lane == 0 && wgid < kStampSlots) g_gemm_timeline[4 * wgid + wave] = stamp_pc - stamp_t0;
    } else if constexpr (ABL & 128) {
      // Where and when the workgroup ran: 100 MHz real-time at its first instruction and behind its last store, and the
      // HW_ID register (CU, shader engine, XCC) of wave 0, to line up the workgroups that followed one another on a CU.
      const unsigned long long r4 = __builtin_amdgcn_s_memrealtime();
      const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID, 32 bits
      const unsigned xcc = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11));   // HW_REG_XCC_ID, 4 bits
      if (threadIdx.x == 0 && wgid < kStampSlots) {
        g_gemm_timeline[4 * wgid] = stamp_r0;
        g_gemm_timeline[4 * wgid + 1] = r4;
        g_gemm_timeline[4 * wgid + 2] = ((unsigned long long)xcc << 32) | hw;
        g_gemm_timeline[4 * wgid + 3] = s4 - stamp_t0;
      }
    } else if constexpr ((ABL & 64) != 0) {
      if (threadIdx.x == 0 && wgid < kStampSlots) {  // the prologue in four steps
        g_gemm_timeline[4 * wgid] = stamp_pa - stamp_t0;
        g_gemm_timeline[4 * wgid + 1] = stamp_pb - stamp_t0;
        g_gemm_timeline[4 * wgid + 2] = stamp_pc - stamp_t0;
        g_gemm_timeline[4 * wgid + 3] = stamp_s1 - stamp_t0;
      }
    } else if (threadIdx.x == 0 && wgid < kStampSlots) {
      g_gemm_timeline[4 * wgid] = stamp_s1 - stamp_t0;
      g_gemm_timeline[4 * wgid + 1] = stamp_s2 - stamp_t0;
      g_gemm_timeline[4 * wgid + 2] = stamp_s3 - stamp_t0;
      g_gemm_timeline[4 * wgid + 3] = s4 - stamp_t0;
    }
  }
#endif
}

// ---- output checksum (integer, order-independent -> bitwise reproducible) ----
using csum_u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ unsigned long long checksum_word(csum_u32x4 v) {
  return (unsigned long long)v.x * 0x9E3779B1u + (unsigned long long)v.y * 0x85EBCA77u + (unsigned long long)v.z * 0xC2B2AE3Du +
         (unsigned long long)v.w * 0x27D4EB2Fu;
}

// Runs between the launches of a soak, so it is written to take HBM time and no more: four independent streaming
// 16-byte loads in flight per lane (C was stored non-temporally and comes from memory, not from a cache).
constexpr int kChecksumSlots = 8;  // words of a result, 64 bytes apart
__global__ __launch_bounds__(256) void checksum_kernel(const uint4* __restrict__ c4, uint64_t n16, unsigned long long* out) {
  const csum_u32x4* c = reinterpret_cast<const csum_u32x4*>(c4);
  unsigned long long s = 0;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const csum_u32x4 v0 = __builtin_nontemporal_load(c + i), v1 = __builtin_nontemporal_load(c + i + stride);
    const csum_u32x4 v2 = __builtin_nontemporal_load(c + i + 2 * stride), v3 = __builtin_nontemporal_load(c + i + 3 * stride);
    s += checksum_word(v0) + checksum_word(v1) + checksum_word(v2) + checksum_word(v3);
  }
  for (; i < n16; i += stride) s += checksum_word(__builtin_nontemporal_load(c + i));
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  // One atomic per workgroup, into one of kChecksumSlots words of a line each (the host adds them up): thousands of
  // 64-bit atomics on one result word are served one after the other by a single L2 channel and took four times as long
  // as the reads themselves when every wave issued its own; a thousand workgroups' still took 10 us of a 31 us pass.
  __shared__ unsigned long long part[4];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out + (blockIdx.x % kChecksumSlots) * 8, part[0] + part[1] + part[2] + part[3]);
}
// Workgroups of a checksum pass over n16 16-byte words: sixteen loads per lane at least (a small C does not pay for a
// thousand workgroups' launch and atomics), at most four workgroups per CU.
unsigned checksum_grid(uint64_t n16) { return (unsigned)std::min<uint64_t>(1024, std::max<uint64_t>(32, n16 / (256 * 16))); }

// ---- operand fill: counter-based uniform [-1, 1) ------------------------------
__device__ __forceinline__ uint32_t hash32(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return (uint32_t)((x ^ (x >> 31)) >> 16);
}

__global__ void fill_bf16_kernel(__bf16* p, uint64_t n, uint64_t seed) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    p[i] = (__bf16)((float)(hash32(i + seed) & 0xffff) * (2.0f / 65536.0f) - 1.0f);
}

// OCP e4m3 bytes covering a spread of magnitudes and both signs; never NaN (0x7f / 0xff).
__global__ void fill_fp8_kernel(uint8_t* p, uint64_t n, uint64_t seed) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    uint32_t h = hash32(i + seed);
    uint8_t mag = (uint8_t)(0x20 + (h & 0x1f));  // exponent 4..7, mantissa random: |x| in [2^-3, 2)
    p[i] = (uint8_t)(mag | ((h >> 8) & 1 ? 0x80 : 0));
  }
}

// E8M0 block scales within 2^-3 .. 2^3 (row-major [rows][K/32]).
__global__ void fill_e8m0_kernel(uint8_t* p, uint64_t n, uint64_t seed) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) p[i] = (uint8_t)(124 + hash32(i + seed) % 7);
}

// Row-major E8M0 scales [rows][kblocks] -> packed units (see the head of this file): one thread per packed dword.
__global__ void pack_scales_kernel(const uint8_t* __restrict__ nat, uint32_t* __restrict__ packed, int rows, int kblocks) {
  const int ktiles = kblocks / 4;
  const uint64_t n = (uint64_t)(rows / 64) * ktiles * kScaleUnitDwords;
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int d = (int)(i % kScaleUnitDwords), fq = d >> 4, fr = d & 15;
    const uint64_t unit = i / kScaleUnitDwords;
    const int kt = (int)(unit % ktiles), rg = (int)(unit / ktiles);
    uint32_t v = 0;
    for (int c = 0; c < 4; c++) v |= (uint32_t)nat[(size_t)(128 * (rg >> 1) + 8 * fr + c + 4 * (rg & 1)) * kblocks + 4 * kt + fq] << (8 * c);
    packed[i] = v;
  }
}

// ---- independent reference for sampled C entries ---------------------------------
// One wave per sample: the full-K dot product of row i of A and row j of B, decoded element by element and summed
// in double.  Shares nothing with the GEMM kernels (no MFMA, no LDS, no tiling, row-major scales).
__device__ __forceinline__ float e4m3_to_float(uint8_t v) {
  const int e = (v >> 3) & 15, m = v & 7;
  const float mag = e ? ldexpf(1.0f + m * 0.125f, e - 7) : ldexpf((float)m, -9);
  return (v & 0x80) ? -mag : mag;
}
__global__ __launch_bounds__(64) void gemm_reference_kernel(int dtype, const void* __restrict__ A, const void* __restrict__ B,
                                                            const uint8_t* __restrict__ sa, const uint8_t* __restrict__ sb,
                                                            const __bf16* __restrict__ C, int M, int N, int K, uint64_t seed,
                                                            float* __restrict__ ref, float* __restrict__ got) {
  const int s = blockIdx.x, lane = threadIdx.x;
  const int i = (int)(hash32(2ull * s + seed) % (uint32_t)M), j = (int)(hash32(2ull * s + 1 + seed) % (uint32_t)N);
  double sum = 0;
  for (int k = lane; k < K; k += 64) {
    float x, y;
    if (dtype == GMD_DT_BF16) {
      x = (float)((const __bf16*)A)[(size_t)i * K + k];
      y = (float)((const __bf16*)B)[(size_t)j * K + k];
    } else {
      x = e4m3_to_float(((const uint8_t*)A)[(size_t)i * K + k]);
      y = e4m3_to_float(((const uint8_t*)B)[(size_t)j * K + k]);
      if (dtype == GMD_DT_MXFP8) {
        x *= ldexpf(1.0f, (int)sa[(size_t)i * (K / 32) + k / 32] - 127);
        y *= ldexpf(1.0f, (int)sb[(size_t)j * (K / 32) + k / 32] - 127);
      }
    }
    sum += (double)x * (double)y;
  }
  for (int off = 32; off > 0; off >>= 1) sum += __shfl_down(sum, off, 64);
  if (lane == 0) {
    ref[s] = (float)sum;
    got[s] = (float)C[(size_t)i * N + j];
  }
}

// ---- matrix-pipe peak -----------------------------------------------------------
// The accumulate-in-place MFMA is written as inline asm: with the builtin hipcc
// (ROCm 7.2) rotates the eight accumulators of this loop through VGPR copies on
// every iteration, which halves the measured rate.  An MFMA result feeding the
// next MFMA whole as its C operand needs no wait states; the results are read
// by compiler code only after PEAK_DRAIN.
#define PEAK_MFMA(op, acc, a, b) asm volatile(op " %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define PEAK_DRAIN(acc) asm volatile("s_nop 15\n\ts_nop 15" : "+v"(acc))
template <int DT>
__global__ __launch_bounds__(256) void peak_kernel(float* __restrict__ out, int iters, unsigned long long* stamps) {
  const int lane = threadIdx.x & 63;
  unsigned long long t0 = 0, r0 = 0;
  if (threadIdx.x == 0 && stamps) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  float keep = 0.f;
  if constexpr (DT == GMD_DT_BF16) {
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      a[j] = (__bf16)(0.25f + 0.01f * ((lane * 7 + j * 3) % 61) - 0.3f);
      b[j] = (__bf16)(0.5f - 0.013f * ((lane * 5 + j * 11) % 53));
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f32x4{(float)i, (float)lane, 0.5f * i, 1.f};  // distinct chains: nothing to merge
    for (int it = 0; it < iters / 8; it++)  // 64 MFMAs per branch: a taken branch is ~13 cycles in which the wave issues nothing
#pragma unroll
      for (int i8 = 0, i = 0; i8 < 64; i8++, i = i8 & 7) PEAK_MFMA("v_mfma_f32_16x16x32_bf16", acc[i], a, b);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PEAK_DRAIN(acc[i]);
      keep += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
  } else if constexpr (DT == GMD_DT_FP8) {
    // Unscaled fp8 on the K=128 instruction, the one the fp8 GEMM runs on (v_mfma_f32_16x16x32_fp8_fp8 takes 16 cycles for
    // a quarter of its work and measured 2421 TFLOP/s in this loop).
    i32x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      a[j] = 0x38302830 ^ (((lane + j) * 0x01010101) & 0x07070707);
      b[j] = 0x30b830b8 ^ (((lane * 3 + j) * 0x01010101) & 0x07070707);
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f32x4{(float)i, (float)lane, 0.5f * i, 1.f};  // distinct chains: nothing to merge
    for (int it = 0; it < iters / 8; it++)  // 64 MFMAs per branch: a taken branch is ~13 cycles in which the wave issues nothing
#pragma unroll
      for (int i8 = 0, i = 0; i8 < 64; i8++, i = i8 & 7) PEAK_MFMA("v_mfma_f32_16x16x128_f8f6f4", acc[i], a, b);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PEAK_DRAIN(acc[i]);
      keep += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
  } else if constexpr (DT == GMD_DT_MXFP8) {
    i32x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      a[j] = 0x38302830 ^ (((lane + j) * 0x01010101) & 0x07070707);
      b[j] = 0x30b830b8 ^ (((lane * 3 + j) * 0x01010101) & 0x07070707);
    }
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f32x4{(float)i, (float)lane, 0.5f * i, 1.f};  // distinct chains: nothing to merge
    for (int it = 0; it < iters / 8; it++)  // 64 MFMAs per branch: a taken branch is ~13 cycles in which the wave issues nothing
#pragma unroll
      for (int i8 = 0, i = 0; i8 < 64; i8++, i = i8 & 7)
        acc[i] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc[i], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PEAK_DRAIN(acc[i]);
      keep += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
  } else {  // f32-input MFMA
    float a = 0.001f * (lane + 1), b = 0.002f * (64 - lane);
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = f32x4{(float)i, (float)lane, 0.5f * i, 1.f};  // distinct chains: nothing to merge
    for (int it = 0; it < iters / 8; it++)  // 64 MFMAs per branch: a taken branch is ~13 cycles in which the wave issues nothing
#pragma unroll
      for (int i8 = 0, i = 0; i8 < 64; i8++, i = i8 & 7) PEAK_MFMA("v_mfma_f32_16x16x4_f32", acc[i], a, b);
#pragma unroll
    for (int i = 0; i < 8; i++) {
      PEAK_DRAIN(acc[i]);
      keep += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    }
  }
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = keep;
  if (threadIdx.x == 0 && stamps) {
    // keep the stamps behind the MFMA chain
    asm volatile("" ::"v"(keep));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

double peak_flops_per_mfma(int dt) {
  switch (dt) {
    case GMD_DT_BF16: return 2.0 * 16 * 16 * 32;
    case GMD_DT_FP8:
    case GMD_DT_MXFP8: return 2.0 * 16 * 16 * 128;
    default: return 2.0 * 16 * 16 * 4;
  }
}

// Which GEMM schedule runs: -1 = the fastest for the dtype and shape (default); 0 = the simple double-buffered
// loop (A/B baseline; also what odd K-tile counts run); 1 = eight waves of 128 x 64, one workgroup per tile;
// 2 = four waves of 128 x 128; 3 = eight waves of 128 x 64 as one stream per CU over its tiles (gemm_ps_kernel);
// 4 / 5 = schedule 1 with the narrow / the wide but cached epilogue (A/B references for store_acc);
// 6 = four waves of 128 x 128 as one stream per CU over its tiles (gemm_w4_kernel<.., STREAM>);
// 7 = schedule 2 on the K=32 fp8 instruction (unscaled fp8 only, where 2 and 6 run the K=32 sums' products on the K=128
// instruction, which adds them up in another order: an A/B reference; schedule 2 for the other data types).
// 8 = 128 x 128 tiles (gemm_s128_kernel, below); 9 / 10 = small / large tiles split along K (plan_split, further below).
// ---- (2f) small tiles for small problems -----------------------------------------
//
// The 256 x 256 tile of the pipelines above is what a large GEMM wants (one wave holds 128 x 128 of C and reads each
// operand byte from LDS once per 128 columns), but a problem with fewer such tiles than the chip has CUs leaves CUs
// idle: 2048^3 is 64 tiles on 256 CUs.  This kernel computes 128 x 128 tiles with four waves of 64 x 64 over a ring of
// K-tile stages in LDS (A and B, 32 KiB per stage): stage t + NSTAGE - 1 is issued before stage t is computed on, so
// that the memory latency of several K-tiles (a K-tile is only 32 MFMAs per wave here) is in flight at any time;
// one counted vmcnt wait and one barrier per K-tile.  Same LDS image and swizzle as gemm_kernel, same order of the sums (K-tiles in
// order, k-substeps in order), so bf16 results are bit-identical to every other schedule's and fp8 results to the
// K=128 schedules'.  A 64 x 64 wave tile reads 8 fragments per 16 MFMAs, which is all the LDS can deliver, so this is
// no kernel for large problems; the dispatcher uses it where the large tiles cannot fill the chip.
// Any M, any N that is a multiple of 8, whole K-tiles: rows beyond the matrix are fetched from its last row and
// their results are not stored.  The SPLIT instantiation walks a range of the K-tiles and leaves its accumulators in a
// workspace instead of storing C (plan_split() below says when).
constexpr int kS128 = 128;
constexpr int kS128Threads = 256;
constexpr int kS128TileBytes = kS128 * BKB;  // 16 KiB per operand per stage
constexpr int kS128Stages = 5;               // 160 KiB: all of a CU's LDS, four K-tiles in flight
constexpr int kS128LdsBytes = kS128Stages * 2 * kS128TileBytes;

// One launch covers up to two regions of one GEMM (the right and the bottom strip of a matrix whose bulk runs on the
// large tiles): workgroups [0, tiles0) are region 0's tiles, the rest region 1's.  Both share K and C's row stride.
struct S128Region {
  const char *a, *b;  // first row of the region's A rows / B rows (B is [N, K])
  __bf16* c;          // the region's first element of C
  int m, n;
};
struct S128Args {
  S128Region r[2];
  int tiles0;
  int ldc;
  size_t row_bytes;
  // Split along K (SPLIT instantiation, region 0 only): workgroup b computes K-tiles [s * kt_per_split, ...) of tile
  // b % tiles0 with s = b / tiles0 and leaves its fp32 accumulators in `work` for gemm_s128_reduce_kernel.
  float* work;
  int kt_per_split;
};

// A wave's accumulators in the split workspace: one 16-byte word per lane and 16 x 16 block, so that a wave's store or
// load of a block is one contiguous KiB.  Slab (split, tile, wave) holds 16 blocks (mb * 4 + nb).  Plain stores and
// loads: the partial sums are read back within microseconds, and non-temporal ones were 1 to 6 % slower per GEMM.
constexpr size_t kS128SlabFloats = 16 * 64 * 4;  // per wave
__device__ __forceinline__ f32x4* s128_slab(float* work, int split, int tiles, int tile, int wave) {
  return (f32x4*)(work + (((size_t)split * tiles + tile) * 4 + wave) * kS128SlabFloats);
}

// The four 16 x 16 blocks of one 16-row block row of a wave's 64 x 64, rounded to bf16 and stored.  B was the MFMA's
// first operand, so lane (fr, fq) holds, per block, row fr and the four columns 4 fq .. 4 fq + 3.  Neighbouring
// 16-lane rows of two adjacent blocks swap halves (v_permlane16_swap), after which every lane holds eight consecutive
// columns of one block: 16-byte stores, each behind its bounds test.  `row` is the lane's row, `col0` the column of
// its eight in the first pair of blocks.
__device__ __forceinline__ unsigned s128_store_blocks(const f32x4 (&acc)[4], __bf16* __restrict__ C, int M, int N, int ldc, int row, int col0) {
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  unsigned sum = 0;  // of the 32-bit words stored (the soak's check of every launch; unused elsewhere)
#pragma unroll
  for (int nb = 0; nb < 4; nb += 2) {
    const u32x2 x = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[nb], bf16x4));
    const u32x2 y = __builtin_bit_cast(u32x2, __builtin_convertvector(acc[nb + 1], bf16x4));
    const u32x2 r0 = __builtin_amdgcn_permlane16_swap(x.x, y.x, false, false);
    const u32x2 r1 = __builtin_amdgcn_permlane16_swap(x.y, y.y, false, false);
    const u32x4 o = u32x4{r0.x, r1.x, r0.y, r1.y};
    const int col = col0 + nb * 16;
    if (row < M && col < N) {  // N is a multiple of 8
      __builtin_nontemporal_store(o, (u32x4*)(C + (size_t)row * ldc + col));
      sum += (o[0] + o[1]) + (o[2] + o[3]);
    }
  }
  return sum;
}

template <int DT, int NSTAGE = kS128Stages, bool SPLIT = false, bool SUM = false>  // SUM: the soak's instantiation (g_gemm_reg_sum)
__global__ __launch_bounds__(kS128Threads, 1) void gemm_s128_kernel(S128Args args) {
  const int region = !SPLIT && (int)blockIdx.x >= args.tiles0 ? 1 : 0;  // workgroup-uniform
  const char* __restrict__ A = args.r[region].a;
  const char* __restrict__ B = args.r[region].b;
  __bf16* __restrict__ C = args.r[region].c;
  const int M = args.r[region].m, N = args.r[region].n, ldc = args.ldc;
  const size_t row_bytes = args.row_bytes;
  static_assert(DT == GMD_DT_BF16 || DT == GMD_DT_FP8, "bf16 and unscaled fp8");
  static_assert(NSTAGE >= 3 && NSTAGE <= 5, "stage t + 1 is read while stage t is computed on; the wait ladder knows up to three later stages");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wr = wave >> 1, wc = wave & 1;

  // XCD-aware tile assignment (bijective for any grid size), tiles walked in groups that share operand panels in L2.
  const int tiles_m = (M + kS128 - 1) / kS128, tiles_n = (N + kS128 - 1) / kS128;
  const int nwg = region ? (int)gridDim.x - args.tiles0 : args.tiles0;
  int split = 0, bid = (int)blockIdx.x - (region ? args.tiles0 : 0);
  bool split_by_xcd = false;
  if constexpr (SPLIT) split_by_xcd = split_of_block((int)blockIdx.x, (int)gridDim.x / args.tiles0, args.tiles0, args.kt_per_split, 32, &split, &bid);
  const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
  const int wgid = split_by_xcd ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
  int tm, tn;
  tile_of(wgid, tiles_m, tiles_n, &tm, &tn);
  const int rows_a = min(M - tm * kS128, kS128), rows_b = min(N - tn * kS128, kS128);
  const int kt0 = SPLIT ? split * args.kt_per_split : 0;  // this workgroup's first K-tile
  const char* gA = A + (size_t)tm * kS128 * row_bytes + (size_t)kt0 * BKB;
  const char* gB = B + (size_t)tn * kS128 * row_bytes + (size_t)kt0 * BKB;

  // Staging: 16 pieces of 8 rows per operand tile; wave w issues pieces 4w..4w+3 of A and of B.  Lane i of a piece
  // lands at piece_base + 16 i and fetches row i >> 3, source chunk (i & 7) ^ f(i >> 3).
  const int r8 = lane >> 3;
  const int src_chunk = ((lane & 7) ^ chunk_swizzle<DT>(r8)) * 16;
  size_t src_a[4], src_b[4];
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int row = (wave * 4 + i) * 8 + r8;
    src_a[i] = (size_t)min(row, rows_a - 1) * row_bytes + src_chunk;
    src_b[i] = (size_t)min(row, rows_b - 1) * row_bytes + src_chunk;
  }
  // LDS-DMA instruction j (0..7) of a stage: piece j >> 1 of this wave's four, A (even j) or B (odd j).
  auto stage_one = [&](int j, int kt, char* buf) {
    const int i = j >> 1, piece = wave * 4 + i;
    if (j & 1)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gB + src_b[i] + (size_t)kt * BKB),
                                       (__attribute__((address_space(3))) void*)(buf + kS128TileBytes + piece * 1024), 16, 0, 0);
    else
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gA + src_a[i] + (size_t)kt * BKB),
                                       (__attribute__((address_space(3))) void*)(buf + piece * 1024), 16, 0, 0);
  };
  auto stage_both = [&](int kt, char* buf) {
#pragma unroll
    for (int j = 0; j < 8; j++) stage_one(j, kt, buf);
  };

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // Fragments: lane (fr, fq) reads row fr of a 16-row block, chunks fq and 4 + fq (bf16: the two k-substeps; fp8: the
  // two halves of the K=128 operand).
  const int fr = lane & 15, fq = lane >> 4;
  int a_off[2], b_off[2];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    a_off[h] = lds_off<DT>(wr * 64 + fr, 4 * h + fq);
    b_off[h] = lds_off<DT>(wc * 64 + fr, 4 * h + fq);
  }

  const int nt = SPLIT ? min(args.kt_per_split, (int)(row_bytes / BKB) - kt0) : (int)(row_bytes / BKB);
  // The fragments of a whole K-tile: eight 16-byte reads per operand (four 16-row blocks x two 64-byte halves).
  struct Frags {
    i32x4 a[4][2], b[4][2];
  };
  // Fragment read r (0..15) of a stage: B blocks first (r 0..7: block r >> 1, half r & 1), then A.
  auto read_one = [&](Frags& f, int r, const char* la) {
    const int blk = (r & 7) >> 1, h = r & 1;
    if (r < 8) f.b[blk][h] = *(const i32x4*)(la + kS128TileBytes + b_off[h] + blk * 2048);
    else f.a[blk][h] = *(const i32x4*)(la + a_off[h] + blk * 2048);
  };
  auto read_frags = [&](Frags& f, int ring_slot) {
    const char* la = smem + ring_slot * 2 * kS128TileBytes;
#pragma unroll
    for (int r = 0; r < 16; r++) read_one(f, r, la);
  };
  auto mfmas = [&](const Frags& f) {
    if constexpr (DT == GMD_DT_BF16) {
#pragma unroll
      for (int ks = 0; ks < 2; ks++)
#pragma unroll
        for (int mb = 0; mb < 4; mb++)
#pragma unroll
          for (int nb = 0; nb < 4; nb++)
            acc[mb][nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, f.b[nb][ks]), __builtin_bit_cast(bf16x8, f.a[mb][ks]),
                                                                  acc[mb][nb], 0, 0, 0);
    } else {
#pragma unroll
      for (int mb = 0; mb < 4; mb++)
#pragma unroll
        for (int nb = 0; nb < 4; nb++) {  // unscaled: both scales 2^0 (E8M0 127)
          const i32x4 bl = f.b[nb][0], bh = f.b[nb][1], al = f.a[mb][0], ah = f.a[mb][1];
          acc[mb][nb] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(i32x8{bl.x, bl.y, bl.z, bl.w, bh.x, bh.y, bh.z, bh.w},
                                                                         i32x8{al.x, al.y, al.z, al.w, ah.x, ah.y, ah.z, ah.w}, acc[mb][nb], 0, 0, 0,
                                                                         0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
    }
  };
  // Stage j has landed once at most `later` of the stages issued after it are outstanding (eight LDS-DMA instructions
  // per wave and stage).
  auto wait_stage = [&](int later) {
    if (later >= 3) wait_vmcnt<24>();
    else if (later == 2) wait_vmcnt<16>();
    else if (later == 1) wait_vmcnt<8>();
    else wait_vmcnt<0>();
  };
  // One K-tile: make sure stage t + 1 has landed everywhere (and that every wave has left stage t - 1 behind, whose
  // slot is filled next), then the eight LDS-DMA instructions of stage t + NSTAGE - 1 and the 16 fragment reads of
  // stage t + 1 in the shadow of the MFMAs of stage t.  With one wave per SIMD nothing else hides an LDS read or the
  // issue of an LDS-DMA instruction, so they are dealt out between the MFMAs: the source states the memory
  // instructions in the order wanted (the compiler will not reorder an LDS-DMA write against an LDS read) and the
  // scheduler is told where the MFMAs go.  STEADY: stage t + NSTAGE - 1 exists, no branch in the body.
  auto ktile = [&](auto STEADY, int t, const Frags& cur, Frags& nxt) {
    constexpr bool kSteady = decltype(STEADY)::value;
    const bool more = kSteady || t + 1 < nt;
    if constexpr (kSteady) wait_vmcnt<8 * (NSTAGE - 3)>();
    else if (more) wait_stage(min(NSTAGE - 3, nt - 2 - t));
    GMD_PHASE_FENCE();
    if constexpr (kSteady) {
      char* const fill = smem + ((t + NSTAGE - 1) % NSTAGE) * 2 * kS128TileBytes;
      const char* const la = smem + ((t + 1) % NSTAGE) * 2 * kS128TileBytes;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        stage_one(j, t + NSTAGE - 1, fill);
        read_one(nxt, 2 * j, la);
        read_one(nxt, 2 * j + 1, la);
      }
      mfmas(cur);
      constexpr int kPerGroup = DT == GMD_DT_BF16 ? 4 : 2;  // MFMAs per LDS-DMA instruction and pair of reads
#pragma unroll
      for (int j = 0; j < 8; j++) {
        __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);          // one LDS-DMA instruction
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // two fragment reads
        __builtin_amdgcn_sched_group_barrier(0x008, kPerGroup, 0);  // MFMAs
      }
    } else {
      if (t + NSTAGE - 1 < nt) stage_both(t + NSTAGE - 1, smem + ((t + NSTAGE - 1) % NSTAGE) * 2 * kS128TileBytes);
      if (more) read_frags(nxt, (t + 1) % NSTAGE);
      mfmas(cur);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // Prologue: stages 0 .. NSTAGE-2 in flight, stage 0 landed and in registers.
#pragma unroll
  for (int j = 0; j < NSTAGE - 1; j++)
    if (j < nt) stage_both(j, smem + j * 2 * kS128TileBytes);
  wait_stage(min(NSTAGE - 2, nt - 1));
  GMD_PHASE_FENCE();
  Frags f0, f1;
  read_frags(f0, 0);
  using Steady = std::integral_constant<bool, true>;
  using Tail = std::integral_constant<bool, false>;
  int t = 0;
  for (; t + NSTAGE < nt; t += 2) {  // both K-tiles of the pair have a stage to issue
    ktile(Steady{}, t, f0, f1);
    ktile(Steady{}, t + 1, f1, f0);
  }
  for (; t + 1 < nt; t += 2) {
    ktile(Tail{}, t, f0, f1);
    ktile(Tail{}, t + 1, f1, f0);
  }
  if (t < nt) ktile(Tail{}, t, f0, f1);

  if constexpr (SPLIT) {  // the sums over this workgroup's K-tiles, as they are
    f32x4* const slab = s128_slab(args.work, split, args.tiles0, tm * tiles_n + tn, wave);
#pragma unroll
    for (int mb = 0; mb < 4; mb++)
#pragma unroll
      for (int nb = 0; nb < 4; nb++) slab[(mb * 4 + nb) * 64 + lane] = acc[mb][nb];
    return;
  }
  const int row0 = tm * kS128 + wr * 64 + fr;
  const int col0 = tn * kS128 + wc * 64 + (fq & 1) * 16 + (fq >> 1) * 8;
  unsigned sum = 0;
#pragma unroll
  for (int mb = 0; mb < 4; mb++) sum += s128_store_blocks(acc[mb], C, M, N, ldc, row0 + mb * 16, col0);
  if constexpr (SUM) reg_sum_add(sum, lane);
}

// Second half of a GEMM split along K: workgroup (tile, w) adds up, for wave w of the tile, the accumulators that the
// `splits` workgroups of the tile left in the workspace, always in the order of the splits (the result does not
// depend on which workgroup finished first), and stores the tile's quarter as the unsplit kernel would.  Wave v of
// the workgroup takes block row v.
template <bool SUM>  // SUM: the soak's instantiation, which adds the words it stores to g_gemm_reg_sum
__global__ __launch_bounds__(kS128Threads) void gemm_s128_reduce_kernel(const float* __restrict__ work, int splits, int tiles, __bf16* __restrict__ C,
                                                                        int M, int N, int ldc) {
  const int lane = threadIdx.x & 63, mb = threadIdx.x >> 6;
  const int tile = (int)blockIdx.x >> 2, wave = (int)blockIdx.x & 3;
  const int tiles_n = (N + kS128 - 1) / kS128;
  const int tm = tile / tiles_n, tn = tile % tiles_n;
  const int wr = wave >> 1, wc = wave & 1, fr = lane & 15, fq = lane >> 4;
  f32x4 acc[4];
#pragma unroll
  for (int nb = 0; nb < 4; nb++) acc[nb] = s128_slab((float*)work, 0, tiles, tile, wave)[(mb * 4 + nb) * 64 + lane];
  for (int s = 1; s < splits; s++) {
    const f32x4* slab = s128_slab((float*)work, s, tiles, tile, wave);
#pragma unroll
    for (int nb = 0; nb < 4; nb++) acc[nb] += slab[(mb * 4 + nb) * 64 + lane];
  }
  const unsigned sum = s128_store_blocks(acc, C, M, N, ldc, tm * kS128 + wr * 64 + fr + mb * 16, tn * kS128 + wc * 64 + (fq & 1) * 16 + (fq >> 1) * 8);
  if constexpr (SUM) reg_sum_add(sum, lane);
}

constexpr int kVariantUnset = -2;
constexpr int kVariantMax = 10;
int g_gemm_variant = kVariantUnset;

int gemm_variant() {
  if (g_gemm_variant == kVariantUnset) {
    const char* e = getenv("GMD_GEMM_VARIANT");
    g_gemm_variant = e && *e ? atoi(e) : -1;
  }
  return g_gemm_variant;
}

// Workgroups of the persistent schedule: one per CU of the current device (128 KiB of LDS each, so the
// dispatcher cannot place two on a CU), a multiple of 8 so every XCD gets the same number.
int persistent_grid() {
  static int grid = 0;
  if (!grid) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
        cus < 8)
      cus = gmd::kCUs;
    grid = cus & ~7;
  }
  return grid;
}

// Launch one instantiation with its dynamic LDS (128 KiB operand ring; MX-fp8 adds 8 KiB of scale ring).  The
// opt-in attribute is per function and device; it is set again whenever the (kernel, device) pair differs from the
// calling thread's previous launch.
typedef void (*gemm_fn_t)(const char*, const char*, __bf16*, int, int, size_t, const uint32_t*, const uint32_t*, int);
struct GemmArgs {
  const void *a, *b;
  void* c;
  int m, n;
  size_t row_bytes;
  const uint32_t *sa, *sb;  // packed MX scales (MX-fp8 only)
  hipStream_t st;
  int ldc = 0;  // C's row stride in elements; 0 = n
  bool reg_sum = false;  // the instantiation whose epilogue sums what it stores into g_gemm_reg_sum (reg_sum_shape())
};
int launch_lds(gemm_fn_t kfn, int grid, int threads, int lds_bytes, const GemmArgs& g) {
  // (a decomposed GEMM alternates between two kernels: the last two pairs are remembered)
  thread_local gemm_fn_t last_fn[2] = {nullptr, nullptr};
  thread_local int last_dev[2] = {-1, -1};
  int dev = 0;
  GMD_HIP(hipGetDevice(&dev));
  if (!((last_fn[0] == kfn && last_dev[0] == dev) || (last_fn[1] == kfn && last_dev[1] == dev))) {
    GMD_HIP(hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    last_fn[1] = last_fn[0], last_dev[1] = last_dev[0];
    last_fn[0] = kfn, last_dev[0] = dev;
  }
  kfn<<<grid, threads, lds_bytes, g.st>>>((const char*)g.a, (const char*)g.b, (__bf16*)g.c, g.m, g.n, g.row_bytes, g.sa, g.sb,
                                          g.ldc ? g.ldc : g.n);
  GMD_HIP(hipGetLastError());
  return GMD_OK;
}

// Split along K where the tiles cannot fill the chip and K is long (1024 x 1024 x 4096 is 64 tiles of 128 x 128 on 256
// CUs, each walking 64 K-tiles; 2048 x 2048 x 8192 is one LDS- and L2-bound small tile per CU, or 64 large tiles): several
// workgroups per tile, each over a range of K-tiles, and a second kernel that adds their fp32 accumulators up in a fixed
// order.  The sums differ from the unsplit order in the last bits, but not from launch to launch.
//   schedule 9:  gemm_s128_kernel<.., SPLIT> + gemm_s128_reduce_kernel, 64 KiB of partial sums per workgroup;
//   schedule 10: gemm_w4_kernel<.., EPI = 4> + gemm_w4_reduce_kernel, 256 KiB per workgroup, whole 256 x 256 tiles and
//                K-tiles in pairs.  The large tile needs half the operand bytes per FLOP of the small one, which is what
//                bounds the small-tile kernel (an XCD's L2 delivers about 70 GB/s to a CU).
// Which of them, and into how many ranges, is decided by a model of the time each takes, in microseconds, fitted to
// scripts/gpu_gemm_split_sweep.py (profiles/gemm_split_sweep.json; bf16 and fp8 cost the same per 128-byte K-tile):
//   a K-tile costs a workgroup 0.475 us on small tiles and 0.9 us on large ones, more with more of the chip busy (the
//   clock under the power cap, and the L2); a launch costs 4 to 6 us before its first and after its last K-tile; the
//   partial sums cost 0.29 us per MiB (written, then read) and the second launch 2 to 3.3 us.
// A split is taken when the model has it 3 % ahead of the unsplit launch.
constexpr int kSplitMaxSlabs = 256;  // workgroups of a split launch (the workspace holds 256 KiB for each)
constexpr int kSplitMax = 8;  // more was measured: 16 ranges are slower wherever they fit (512 x 512 x 8192: 18.2 -> 20.3 us)
struct SplitPlan {
  int schedule;  // 0: no split; 9: small tiles; 10: large tiles
  int splits;
  int per;       // K-tiles per range (the last range: what is left)
};
int g_gemm_splits = 0;  // gmd_gemm_set_split: the number of ranges schedules 9 and 10 use when asked for; 0 = their own choice

SplitPlan small_split(int tiles, int ktiles, int s) {
  const int room = std::min(persistent_grid(), kSplitMaxSlabs);
  if (tiles <= 0 || s < 2 || s > kSplitMax || tiles * s > room || s > ktiles) return SplitPlan{0, 1, ktiles};
  const int per = (ktiles + s - 1) / s, eff = (ktiles + per - 1) / per;  // no range without a K-tile
  return eff >= 2 ? SplitPlan{9, eff, per} : SplitPlan{0, 1, ktiles};
}
SplitPlan large_split(int m, int n, int ktiles, int s) {
  const int room = std::min(persistent_grid(), kSplitMaxSlabs);
  if (m % BM || n % BN || ktiles < 4 || (ktiles & 1)) return SplitPlan{0, 1, ktiles};
  const int tiles = (m / BM) * (n / BN);
  if (s < 2 || s > kSplitMax || tiles * s > room || 2 * s > ktiles) return SplitPlan{0, 1, ktiles};
  const int per = ((ktiles + s - 1) / s + 1) & ~1, eff = (ktiles + per - 1) / per;  // whole pairs of K-tiles
  return eff >= 2 ? SplitPlan{10, eff, per} : SplitPlan{0, 1, ktiles};
}

SplitPlan plan_split(int m, int n, int ktiles, int asked, bool large_only = false) {  // large_only: MX-fp8 has no small-tile kernel
  const int cus = persistent_grid();
  const int t128 = large_only ? 0 : ((m + kS128 - 1) / kS128) * ((n + kS128 - 1) / kS128);
  const bool whole = m % BM == 0 && n % BN == 0 && ktiles >= 2 && (ktiles & 1) == 0;
  const int large = whole ? (m / BM) * (n / BN) : 0;
  const SplitPlan none{0, 1, ktiles};
  if (large_only && asked == 9) return none;
  if (asked == 9 || asked == 10) {  // the largest number of ranges that fits, or the number set
    for (int s = g_gemm_splits ? g_gemm_splits : kSplitMax; s >= 2; s--) {
      const SplitPlan p = asked == 9 ? small_split(t128, ktiles, s) : large_split(m, n, ktiles, s);
      if (p.schedule || g_gemm_splits) return p;
    }
    return none;
  }
  const double full = (double)cus;
  auto busy = [&](int wgs) { return std::min(wgs, cus) / full; };
  auto rounds = [&](int wgs) { return (double)((wgs + cus - 1) / cus); };
  auto small_ktile = [&](int wgs) { return 0.475 * (1.0 + 0.15 * busy(wgs) * busy(wgs)); };
  auto large_ktile = [&](int wgs) { return 0.9 + 0.35 * busy(wgs) * busy(wgs) * busy(wgs); };
  // what runs unsplit: small tiles while every CU gets at most one, large tiles beyond (launch_gemm_t)
  double best = t128 > 0 && t128 <= cus ? 4.0 + ktiles * small_ktile(t128) : large ? 5.5 + rounds(large) * ktiles * 0.89 : 0.0;
  if (best == 0.0) return none;  // more small tiles than CUs and no whole large tiles: the general kernel
  best *= 0.97;
  SplitPlan plan = none;
  for (int s = 2; s <= kSplitMax; s++) {
    const SplitPlan p = small_split(t128, ktiles, s);
    if (!p.schedule || p.per < 4) continue;
    const int wgs = t128 * p.splits;
    const double t = 7.3 + p.per * small_ktile(wgs) + 0.29 * wgs / 16.0;
    if (t < best) best = t, plan = p;
  }
  for (int s = 2; s <= kSplitMax; s++) {
    const SplitPlan p = large_split(m, n, ktiles, s);
    if (!p.schedule || p.per < 16) continue;
    const int wgs = large * p.splits;
    const double t = 6.0 + p.per * large_ktile(wgs) + 0.29 * wgs / 4.0;
    if (t < best) best = t, plan = p;
  }
  return plan;
}

// What the split kernels' indexing rests on, checked once more at the launch: every range has K-tiles (whole pairs on large
// tiles), and every workgroup has its slab in the workspace.  A plan that fails this is not launched (the GEMM runs unsplit).
bool split_plan_fits(const SplitPlan& p, int tiles, int ktiles) {
  if (p.splits < 2 || p.per < 1 || tiles < 1 || (long long)tiles * p.splits > kSplitMaxSlabs) return false;
  if ((long long)(p.splits - 1) * p.per >= ktiles || (long long)p.splits * p.per < ktiles) return false;
  if (p.schedule == 10) return (p.per & 1) == 0 && (ktiles & 1) == 0;
  return p.schedule == 9;
}

// The workspace of the split: one per device and stream (launches on one stream follow one another; two streams may
// run two split GEMMs at once), allocated at first use and kept for the life of the process.
float* split_workspace(hipStream_t st) {
  struct Ws {
    int dev;
    hipStream_t st;
    float* p;
  };
  static std::mutex mu;
  static std::vector<Ws> all;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  int held = 0;
  for (const Ws& w : all) {
    if (w.dev == dev && w.st == st) return w.p;
    held += w.dev == dev && w.p;
  }
  // Eight streams of a device get a workspace (64 MiB each); a process that goes on making streams runs its further
  // GEMMs unsplit rather than growing without bound (a workspace is never taken back: a launch may still be using it).
  if (held >= 8) return nullptr;
  float* p = nullptr;
  static_assert(kW4SlabFloats >= kS128SlabFloats, "sized for the large tiles: 256 workgroups x 256 KiB");
  if (hipMalloc((void**)&p, (size_t)kSplitMaxSlabs * 4 * kW4SlabFloats * sizeof(float)) != hipSuccess) {
    (void)hipGetLastError();
    p = nullptr;  // remembered: the caller runs unsplit, now and later
  }
  all.push_back(Ws{dev, st, p});
  return p;
}

constexpr int kScaleRingBytes = 4 * 8 * 256;  // gemm_w4_kernel, MX-fp8: four waves x eight slots x one unit

template <int DT>
int launch_gemm_t(const GemmArgs& g) {
  constexpr bool kMx = DT == GMD_DT_MXFP8;
  const bool full_tiles = g.m % BM == 0 && g.n % BN == 0;
  const int grid = ((g.m + BM - 1) / BM) * ((g.n + BN - 1) / BN);
  const size_t ktiles = g.row_bytes / BKB;
  const bool pairs = ktiles >= 2 && (ktiles & 1) == 0;                      // the pipelines walk K-tiles in pairs
  const bool stream_ok = full_tiles && pairs && ktiles >= 4 && grid > persistent_grid();  // more tiles than CUs, two pairs per tile
  // Partial tiles or an odd number of K-tiles: the general instantiation of the four-wave kernel, whatever was asked for
  // (schedule 0 takes any K-tile count as well, but full tiles only).
  // Unscaled fp8 runs on the K=128 instruction (the MX-fp8 kernel without scales, UNIT); schedule 7 asks for the four-wave
  // kernel on the K=32 instruction instead, the form whose sums agree bit for bit with the eight-wave schedules.
  // The K=128 instruction adds the products up in another order, so the schedules that are A/B references for one another
  // (0, 1, 3, 4, 5 and 7) all stay on the K=32 one, for every shape.
  constexpr bool kFp8 = DT == GMD_DT_FP8;
  const int asked = gemm_variant();
  const bool fp8_wide = kFp8 && (asked < 0 || asked > kVariantMax || asked == 2 || asked == 6 || asked == 8 || asked == 9 || asked == 10);
  // Small problems and ragged edges (bf16 and unscaled fp8; MX-fp8 has no small-tile kernel).
  // * Where the 256 x 256 tiles leave CUs idle, 128 x 128 tiles (schedule 8) fill them: taken when every CU gets at
  //   most one small tile (2048^3: 64 large tiles on 256 CUs, or 256 small ones).  With several small tiles per CU the
  //   kernel is bound by L2 bandwidth (a 128 x 128 tile has half the arithmetic intensity) and the large tiles win.
  // * A matrix a little wider or taller than a whole number of large tiles costs a whole extra column or row of them
  //   (4104 columns are 17 tile columns, the last 8 columns wide), and with it often a whole extra round over the
  //   CUs.  The part that large tiles cover exactly then runs on them and the strips on small tiles: C's row stride
  //   (ldc) stays the full width.  Taken when the model says it saves at least a tenth: a round of large tiles is 1,
  //   a round of small ones 1 / (4 x 0.57) (measured at 8192^3).
  // Schedule 8 asks for small tiles whatever the shape.
  if constexpr (kMx) {  // MX-fp8: large tiles only, split along K where the model has it ahead (or schedule 10 asks)
    const bool automatic = asked < 0 || asked > kVariantMax;
    if ((asked == 10 || automatic) && !g.ldc && full_tiles) {
      const SplitPlan plan = plan_split(g.m, g.n, (int)ktiles, automatic ? -1 : 10, true);
      float* const work = plan.schedule == 10 && split_plan_fits(plan, grid, (int)ktiles) ? split_workspace(g.st) : nullptr;
      if (work) {
        GemmArgs part = g;
        part.c = work;
        part.ldc = plan.per;
        const int rc = launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 4>, grid * plan.splits, kThreadsW4, kLdsBytes + kScaleRingBytes, part);
        if (rc) return rc;
        if (g.reg_sum) gemm_w4_reduce_kernel<true><<<grid * 8, 256, 0, g.st>>>(work, plan.splits, grid, g.n / BN, (__bf16*)g.c, (size_t)g.n);
        else gemm_w4_reduce_kernel<false><<<grid * 8, 256, 0, g.st>>>(work, plan.splits, grid, g.n / BN, (__bf16*)g.c, (size_t)g.n);
        GMD_HIP(hipGetLastError());
        return GMD_OK;
      }
    }
  }
  if constexpr (!kMx) {
    const int cus = persistent_grid();
    auto tiles128 = [](int m, int n) { return ((m + kS128 - 1) / kS128) * ((n + kS128 - 1) / kS128); };
    const int ldc = g.ldc ? g.ldc : g.n;
    auto small = [&](const S128Region& r0, const S128Region& r1) -> int {
      S128Args a;
      a.r[0] = r0, a.r[1] = r1;
      a.tiles0 = tiles128(r0.m, r0.n);
      a.ldc = ldc;
      a.row_bytes = g.row_bytes;
      a.work = nullptr, a.kt_per_split = 0;
      const int grid_s = a.tiles0 + (r1.m > 0 && r1.n > 0 ? tiles128(r1.m, r1.n) : 0);
      thread_local int attr_dev = -1;
      int dev = 0;
      GMD_HIP(hipGetDevice(&dev));
      if (attr_dev != dev) {
        GMD_HIP(hipFuncSetAttribute((const void*)gemm_s128_kernel<DT>, hipFuncAttributeMaxDynamicSharedMemorySize, kS128LdsBytes));
        GMD_HIP(hipFuncSetAttribute((const void*)gemm_s128_kernel<DT, kS128Stages, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    kS128LdsBytes));
        attr_dev = dev;
      }
      if (g.reg_sum) gemm_s128_kernel<DT, kS128Stages, false, true><<<grid_s, kS128Threads, kS128LdsBytes, g.st>>>(a);
      else gemm_s128_kernel<DT><<<grid_s, kS128Threads, kS128LdsBytes, g.st>>>(a);
      GMD_HIP(hipGetLastError());
      return GMD_OK;
    };
    const S128Region whole_matrix{(const char*)g.a, (const char*)g.b, (__bf16*)g.c, g.m, g.n}, none{nullptr, nullptr, nullptr, 0, 0};
    const bool automatic = asked < 0 || asked > kVariantMax;
    // Split along K where the model (plan_split) has it ahead, or where it is asked for.
    if ((asked == 9 || asked == 10 || automatic) && !g.ldc) {
      const SplitPlan plan = plan_split(g.m, g.n, (int)ktiles, automatic ? -1 : asked);
      const bool fits = plan.schedule && split_plan_fits(plan, plan.schedule == 10 ? grid : tiles128(g.m, g.n), (int)ktiles) &&
                        (plan.schedule != 10 || (full_tiles && pairs));
      float* const work = fits ? split_workspace(g.st) : nullptr;
      if (work && plan.schedule == 10) {
        GemmArgs part = g;
        part.c = work;
        part.ldc = plan.per;
        int rc;  // (unscaled fp8 is on the K=128 instruction here: asked for or automatic)
        if constexpr (kFp8) rc = launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 4, false, false, true>, grid * plan.splits, kThreadsW4, kLdsBytes, part);
        else rc = launch_lds(gemm_w4_kernel<DT, 0, 4>, grid * plan.splits, kThreadsW4, kLdsBytes, part);
        if (rc) return rc;
        if (g.reg_sum) gemm_w4_reduce_kernel<true><<<grid * 8, 256, 0, g.st>>>(work, plan.splits, grid, g.n / BN, (__bf16*)g.c, (size_t)g.n);
        else gemm_w4_reduce_kernel<false><<<grid * 8, 256, 0, g.st>>>(work, plan.splits, grid, g.n / BN, (__bf16*)g.c, (size_t)g.n);
        GMD_HIP(hipGetLastError());
        return GMD_OK;
      }
      if (work && plan.schedule == 9) {
        const int tiles = tiles128(g.m, g.n);
        S128Args a;
        a.r[0] = whole_matrix, a.r[1] = none;
        a.tiles0 = tiles;
        a.ldc = ldc;
        a.row_bytes = g.row_bytes;
        a.work = work;
        a.kt_per_split = plan.per;
        thread_local int attr_dev = -1;
        int dev = 0;
        GMD_HIP(hipGetDevice(&dev));
        if (attr_dev != dev) {
          GMD_HIP(hipFuncSetAttribute((const void*)gemm_s128_kernel<DT, kS128Stages, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kS128LdsBytes));
          attr_dev = dev;
        }
        gemm_s128_kernel<DT, kS128Stages, true><<<tiles * plan.splits, kS128Threads, kS128LdsBytes, g.st>>>(a);
        GMD_HIP(hipGetLastError());
        if (g.reg_sum) gemm_s128_reduce_kernel<true><<<tiles * 4, kS128Threads, 0, g.st>>>(work, plan.splits, tiles, (__bf16*)g.c, g.m, g.n, ldc);
        else gemm_s128_reduce_kernel<false><<<tiles * 4, kS128Threads, 0, g.st>>>(work, plan.splits, tiles, (__bf16*)g.c, g.m, g.n, ldc);
        GMD_HIP(hipGetLastError());
        return GMD_OK;
      }
    }
    if (asked == 8 || asked == 9 || (automatic && tiles128(g.m, g.n) <= cus)) return small(whole_matrix, none);
    const int m_main = g.m / BM * BM, n_main = g.n / BN * BN;
    if (automatic && !g.ldc && m_main > 0 && n_main > 0 && (m_main < g.m || n_main < g.n)) {
      auto rounds = [&](int tiles) { return (double)((tiles + cus - 1) / cus); };
      const int strip_tiles = (n_main < g.n ? tiles128(m_main, g.n - n_main) : 0) + (m_main < g.m ? tiles128(g.m - m_main, g.n) : 0);
      const double whole = rounds(grid), split = rounds((m_main / BM) * (n_main / BN)) + rounds(strip_tiles) / (4 * 0.57);
      if (split < 0.9 * whole) {
        GemmArgs main = g;
        main.m = m_main, main.n = n_main, main.ldc = g.n;
        int rc = launch_gemm_t<DT>(main);  // full tiles: the pipelines proper
        if (rc) return rc;
        // right strip: rows [0, m_main), columns [n_main, n); bottom strip with the corner: rows [m_main, m), all columns
        const S128Region right{(const char*)g.a, (const char*)g.b + (size_t)n_main * g.row_bytes, (__bf16*)g.c + n_main, m_main, g.n - n_main};
        const S128Region bottom{(const char*)g.a + (size_t)m_main * g.row_bytes, (const char*)g.b, (__bf16*)g.c + (size_t)m_main * g.n, g.m - m_main, g.n};
        if (n_main < g.n && m_main < g.m) return small(right, bottom);
        return small(n_main < g.n ? right : bottom, none);
      }
    }
  }
  if (g.reg_sum && (!full_tiles || !pairs)) {  // the soak at a ragged shape: the general instantiation with the summing epilogue
    // (unscaled fp8 is on the K=128 instruction here: the sum is taken only with the schedule left to the launcher)
    if constexpr (kFp8) return launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 3, false, true, true>, grid, kThreadsW4, kLdsBytes, g);
    else return launch_lds(gemm_w4_kernel<DT, 0, 3, false, true>, grid, kThreadsW4, kLdsBytes + (kMx ? kScaleRingBytes : 0), g);
  }
  if (!full_tiles || (!pairs && gemm_variant() != 0)) {
    if (fp8_wide)
      return launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 1, false, true, true>, grid, kThreadsW4, kLdsBytes, g);
    return launch_lds(gemm_w4_kernel<DT, 0, 1, false, true>, grid, kThreadsW4, kLdsBytes + (kMx ? kScaleRingBytes : 0), g);
  }
  if (g.reg_sum) {  // the soak: the default kernel of the data type with the summing epilogue (whole tiles, K-tile pairs)
    if constexpr (kFp8) return launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 3, false, false, true>, grid, kThreadsW4, kLdsBytes, g);
    else return launch_lds(gemm_w4_kernel<DT, 0, 3>, grid, kThreadsW4, kLdsBytes + (kMx ? kScaleRingBytes : 0), g);
  }
#define GMD_LAUNCH(KFN, GRID, THREADS) return launch_lds(KFN, GRID, THREADS, kLdsBytes, g)
#ifdef GMD_GEMM_ABLATE
  // Timing-only builds of the per-tile pipeline with parts removed (10 + mask), other issue distances (20 + d),
  // clock stamps (41..46, 51) and the per-tile timeline of the streaming kernel (53).  bf16 and fp8.
  if constexpr (!kMx) {
    switch (gemm_variant()) {
      case 25: GMD_LAUNCH((gemm_pp_kernel<DT, 0, 5>), grid, kThreads);
      case 26: GMD_LAUNCH((gemm_pp_kernel<DT, 0, 6>), grid, kThreads);
      case 27: GMD_LAUNCH((gemm_pp_kernel<DT, 0, 7>), grid, kThreads);
      case 11: GMD_LAUNCH((gemm_pp_kernel<DT, 1>), grid, kThreads);
      case 12: GMD_LAUNCH((gemm_pp_kernel<DT, 2>), grid, kThreads);
      case 13: GMD_LAUNCH((gemm_pp_kernel<DT, 3>), grid, kThreads);
      case 14: GMD_LAUNCH((gemm_pp_kernel<DT, 4>), grid, kThreads);
      case 15: GMD_LAUNCH((gemm_pp_kernel<DT, 5>), grid, kThreads);
      case 16: GMD_LAUNCH((gemm_pp_kernel<DT, 6>), grid, kThreads);
      case 17: GMD_LAUNCH((gemm_pp_kernel<DT, 7>), grid, kThreads);
      case 18: GMD_LAUNCH((gemm_pp_kernel<DT, 8>), grid, kThreads);     // full pipeline, operands always L2-resident
      case 23: GMD_LAUNCH((gemm_pp_kernel<DT, 13>), grid, kThreads);    // DMA only, operands always L2-resident
      case 33: GMD_LAUNCH((gemm_pp_kernel<DT, 5, 5>), grid, kThreads);  // DMA only at issue distance 5
      case 34: GMD_LAUNCH((gemm_pp_kernel<DT, 5, 6>), grid, kThreads);  // DMA only at issue distance 6
      case 41: GMD_LAUNCH((gemm_pp_kernel<DT, 16>), grid, kThreads);
      case 42: GMD_LAUNCH((gemm_pp_kernel<DT, 16 | 3>), grid, kThreads);
      case 43: GMD_LAUNCH((gemm_pp_kernel<DT, 16 | 1>), grid, kThreads);
      case 44: GMD_LAUNCH((gemm_pp_kernel<DT, 16 | 2>), grid, kThreads);
      case 45: GMD_LAUNCH((gemm_pp_kernel<DT, 16 | 5>), grid, kThreads);
      case 46: GMD_LAUNCH((gemm_pp_kernel<DT, 16 | 8>), grid, kThreads);
      case 53: GMD_LAUNCH((gemm_ps_kernel<DT, 1, 16>), persistent_grid(), kThreads);
      default: break;
    }
  }
  {
    const int w4_lds = kLdsBytes + (kMx ? kScaleRingBytes : 0);
    switch (gemm_variant()) {
      case 51: return launch_lds(gemm_w4_kernel<DT, 16>, grid, kThreadsW4, w4_lds, g);
      case 62: return launch_lds(gemm_w4_kernel<DT, 16 | 3>, grid, kThreadsW4, w4_lds, g);   // MFMAs + barriers
      case 63: return launch_lds(gemm_w4_kernel<DT, 16 | 1>, grid, kThreadsW4, w4_lds, g);   // no fragment reads
      case 64: return launch_lds(gemm_w4_kernel<DT, 16 | 2>, grid, kThreadsW4, w4_lds, g);   // no DMA
      case 65: return launch_lds(gemm_w4_kernel<DT, 16 | 5>, grid, kThreadsW4, w4_lds, g);   // DMA only
      case 66: return launch_lds(gemm_w4_kernel<DT, 16 | 32>, grid, kThreadsW4, w4_lds, g);  // no barriers
      case 67: return launch_lds(gemm_w4_kernel<DT, 16 | 3 | 32>, grid, kThreadsW4, w4_lds, g);  // MFMAs alone
      case 68: return launch_lds(gemm_w4_kernel<DT, 16 | (1 << 8)>, grid, kThreadsW4, w4_lds, g);  // pieces back to back (old)
      case 69: return launch_lds(gemm_w4_kernel<DT, 16 | (2 << 8)>, grid, kThreadsW4, w4_lds, g);  // reads 2 apart, pieces 4 apart
      case 70: return launch_lds(gemm_w4_kernel<DT, 16 | (3 << 8)>, grid, kThreadsW4, w4_lds, g);  // pieces 5 apart from 12
      case 71: return launch_lds(gemm_w4_kernel<DT, 16 | (4 << 8)>, grid, kThreadsW4, w4_lds, g);  // reads 2 apart, pieces 3 apart
      case 77: return launch_lds(gemm_w4_kernel<DT, 16 | 8>, grid, kThreadsW4, w4_lds, g);  // stamps; M0 is never written (every half-tile lands in one slot)
      case 78: return launch_lds(gemm_w4_kernel<DT, 16 | 64 | 128>, grid, kThreadsW4, w4_lds, g);  // stamps: the prologue's wait, per wave
      case 76: return launch_lds(gemm_w4_kernel<DT, 16 | 64>, grid, kThreadsW4, w4_lds, g);  // stamps: the prologue in four steps
      case 75: return launch_lds(gemm_w4_kernel<DT, 16 | 128>, grid, kThreadsW4, w4_lds, g);  // stamps: when and on which CU
      case 73: return launch_lds(gemm_w4_kernel<DT, 16, 2>, grid, kThreadsW4, w4_lds, g);  // stamps, cached stores
      case 74: return launch_lds(gemm_w4_kernel<DT, 16, 0>, grid, kThreadsW4, w4_lds, g);  // stamps, narrow cached stores
      case 72:  // the stream with cached stores
        if (stream_ok) return launch_lds(gemm_w4_kernel<DT, 0, 2, true>, persistent_grid(), kThreadsW4, w4_lds, g);
        break;
      default: break;
    }
  }
#endif
  // auto, by measurement (scripts/gpu_gemm_ab.py): the four-wave kernel for every data type and shape it covers.  It is
  // the only pipeline that carries MX scales, and since its epilogue stores whole rows from neighbouring lanes it is
  // ahead of the eight-wave kernels on bf16 and fp8 at 4096 (4-6 %), 8192 (0.5-1.6 %) and 16384 (0.6-1.1 %) alike.
  int variant = gemm_variant();
  if (variant < 0 || variant >= 8) variant = 2;
  if (kMx && variant != 0 && variant != 6) variant = 2;
  if (!pairs) variant = 0;
  if (variant == 3 && !stream_ok) variant = 1;
  if (variant == 6 && !stream_ok) variant = 2;
  if (variant == 7 && !kFp8) variant = 2;
  if (fp8_wide && variant == 2) return launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 1, false, false, true>, grid, kThreadsW4, kLdsBytes, g);
  if (fp8_wide && variant == 6)
    return launch_lds(gemm_w4_kernel<GMD_DT_MXFP8, 0, 1, true, false, true>, persistent_grid(), kThreadsW4, kLdsBytes, g);
  if (variant == 2 || variant == 7) return launch_lds(gemm_w4_kernel<DT>, grid, kThreadsW4, kLdsBytes + (kMx ? kScaleRingBytes : 0), g);
  if (variant == 6)
    return launch_lds(gemm_w4_kernel<DT, 0, 1, true>, persistent_grid(), kThreadsW4, kLdsBytes + (kMx ? kScaleRingBytes : 0), g);
  if constexpr (!kMx) {
    switch (variant) {
      case 1: GMD_LAUNCH((gemm_pp_kernel<DT>), grid, kThreads);
      case 3: GMD_LAUNCH((gemm_ps_kernel<DT>), persistent_grid(), kThreads);
      case 4: GMD_LAUNCH((gemm_pp_kernel<DT, 0, kAheadDefault, 0>), grid, kThreads);
      case 5: GMD_LAUNCH((gemm_pp_kernel<DT, 0, kAheadDefault, 2>), grid, kThreads);
      default: break;
    }
  }
  GMD_LAUNCH((gemm_kernel<DT>), grid, kThreads);
#undef GMD_LAUNCH
}

}  // namespace

extern "C" {

int gmd_gemm_set_variant(int variant) {
#ifdef GMD_GEMM_ABLATE
  if ((variant >= 11 && variant <= 18) || variant == 23 || (variant >= 25 && variant <= 27) || variant == 33 || variant == 34 ||
      (variant >= 41 && variant <= 46) || variant == 51 || variant == 53 || (variant >= 62 && variant <= 78)) {
    g_gemm_variant = variant;
    return GMD_OK;
  }
#endif
  if (variant < -1 || variant > kVariantMax) return GMD_ERR_BADPARAM;
  g_gemm_variant = variant;
  return GMD_OK;
}

// What the schedule in force does with a GEMM of this shape on the current device: *schedule 9 or 10 and *splits ranges
// of K, or *schedule 0 and one range (no split; which unsplit kernel runs is launch_gemm_t's business).
int gmd_gemm_plan(int dtype, int m, int n, int k, int* schedule, int* splits) {
  if (schedule) *schedule = 0;
  if (splits) *splits = 1;
  if ((dtype != GMD_DT_BF16 && dtype != GMD_DT_FP8 && dtype != GMD_DT_MXFP8) || m <= 0 || n <= 0 || k <= 0) return GMD_OK;
  const int asked = gemm_variant();
  const bool automatic = asked < 0 || asked > kVariantMax;
  if (!automatic && asked != 9 && asked != 10) return GMD_OK;
  const SplitPlan p = plan_split(m, n, (int)((size_t)k * (dtype == GMD_DT_BF16 ? 2 : 1) / BKB), automatic ? -1 : asked, dtype == GMD_DT_MXFP8);
  if (schedule) *schedule = p.schedule;
  if (splits) *splits = p.splits;
  return GMD_OK;
}
int gmd_gemm_split_k(int dtype, int m, int n, int k) {
  int splits = 1;
  gmd_gemm_plan(dtype, m, n, k, nullptr, &splits);
  return splits;
}
int gmd_gemm_set_split(int splits) {
  if (splits < 0 || splits == 1 || splits > kSplitMax) return GMD_ERR_BADPARAM;
  g_gemm_splits = splits;
  return GMD_OK;
}

// Ablation builds only: per-workgroup (shader ticks, 100 MHz ticks) pairs of the last stamped launch.
int gmd_gemm_read_stamps(unsigned long long* out, int n_pairs) {
#ifdef GMD_GEMM_ABLATE
  if (!out || n_pairs <= 0 || n_pairs > kStampSlots) return GMD_ERR_BADPARAM;
  GMD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * 2 * n_pairs));
  return GMD_OK;
#else
  (void)out;
  (void)n_pairs;
  return GMD_ERR_UNSUPPORTED;
#endif
}

// Ablation builds only: per-tile timeline of the last stamped persistent launch (variant 53), four values a tile.
int gmd_gemm_read_timeline(unsigned long long* out, int n_tiles) {
#ifdef GMD_GEMM_ABLATE
  if (!out || n_tiles <= 0 || n_tiles > kStampSlots) return GMD_ERR_BADPARAM;
  GMD_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gemm_timeline), sizeof(unsigned long long) * 4 * n_tiles));
  return GMD_OK;
#else
  (void)out;
  (void)n_tiles;
  return GMD_ERR_UNSUPPORTED;
#endif
}

// Whether a launch sums what it stores (the soak's check of every launch): with the schedule left to the launcher every
// path has such an instantiation (large tiles, the general instantiation, small tiles, strips of small tiles next to
// large ones, and the kernels that add up a GEMM split along K); a schedule asked for by number runs as it is.
static bool reg_sum_shape(int, int, int, int) {
  const int asked = gemm_variant();
  return asked < 0 || asked > kVariantMax;
}

static int gemm_launch_checked(int dtype, const void* a, const void* b, const void* sa, const void* sb, void* c, int m, int n, int k,
                               void* hip_stream, bool reg_sum = false) {
  // bf16 and fp8: M and N in multiples of 8 (a lane stores eight columns; rows are dealt to lanes in groups of eight);
  // MX-fp8: whole 256 x 256 tiles (the packed scales come in units of 128 rows and have no notion of a partial one).  K in whole 128-byte K-tiles.
  const int mn_step = dtype == GMD_DT_MXFP8 ? BM : 8;
  if (!a || !b || !c || m <= 0 || n <= 0 || k <= 0 || (m % mn_step) || (n % mn_step) || (k % (dtype == GMD_DT_BF16 ? 64 : 128))) {
    gmd::set_error("GEMM shape: M and N in multiples of 8 (bf16, fp8) or 256 (MX-fp8), K in multiples of 64 (bf16) or 128 (fp8, MX-fp8)");
    return GMD_ERR_BADPARAM;
  }
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c) & 15) {
    gmd::set_error("GEMM operands must be 16-byte aligned");
    return GMD_ERR_BADPARAM;
  }
  GemmArgs g{a, b, c, m, n, 0, (const uint32_t*)sa, (const uint32_t*)sb, (hipStream_t)hip_stream};
  g.reg_sum = reg_sum;
  switch (dtype) {
    case GMD_DT_BF16:
      if (k % 64) return GMD_ERR_BADPARAM;
      g.row_bytes = (size_t)k * 2;
      return launch_gemm_t<GMD_DT_BF16>(g);
    case GMD_DT_FP8:
      if (k % 128) return GMD_ERR_BADPARAM;
      g.row_bytes = (size_t)k;
      return launch_gemm_t<GMD_DT_FP8>(g);
    case GMD_DT_MXFP8:
      if ((k % 128) || !sa || !sb || (((uintptr_t)sa | (uintptr_t)sb) & 3)) {
        gmd::set_error("MX-fp8 GEMM: K must be a multiple of 128 and both packed scale tensors are required (gmd_gemm_launch_mx)");
        return GMD_ERR_BADPARAM;
      }
      g.row_bytes = (size_t)k;
      return launch_gemm_t<GMD_DT_MXFP8>(g);
    default: return GMD_ERR_BADPARAM;
  }
}

int gmd_gemm_launch(int dtype, const void* a, const void* b, void* c, int m, int n, int k, void* hip_stream) {
  return gemm_launch_checked(dtype, a, b, nullptr, nullptr, c, m, n, k, hip_stream);
}

int gmd_gemm_launch_mx(const void* a, const void* b, const void* scale_a_packed, const void* scale_b_packed, void* c, int m, int n,
                       int k, void* hip_stream) {
  return gemm_launch_checked(GMD_DT_MXFP8, a, b, scale_a_packed, scale_b_packed, c, m, n, k, hip_stream);
}

int gmd_mx_pack_scales(const void* scales_row_major, void* packed, int rows, int k, void* hip_stream) {
  if (!scales_row_major || !packed || rows <= 0 || k <= 0 || (rows % 128) || (k % 128) || ((uintptr_t)packed & 3)) return GMD_ERR_BADPARAM;
  pack_scales_kernel<<<1024, 256, 0, (hipStream_t)hip_stream>>>((const uint8_t*)scales_row_major, (uint32_t*)packed, rows, k / 32);
  GMD_HIP(hipGetLastError());
  return GMD_OK;
}

int gmd_gemm_stress(int device, int dtype, int m, int n, int k, double seconds, int verify, gmd_mfma_result_t* out) {
  gmd_gemm_stress_config_t cfg;
  memset(&cfg, 0, sizeof cfg);
  cfg.device = device, cfg.dtype = dtype, cfg.m = m, cfg.n = n, cfg.k = k, cfg.seconds = seconds, cfg.verify = verify;
  return gmd_gemm_stress_run(&cfg, out);
}

int gmd_gemm_stress_run(const gmd_gemm_stress_config_t* cfg, gmd_mfma_result_t* out) {
  if (!out || !cfg) return GMD_ERR_BADPARAM;
  memset(out, 0, sizeof *out);
  const int device = cfg->device, dtype = cfg->dtype, m = cfg->m, n = cfg->n, k = cfg->k, verify = cfg->verify;
  const double seconds = cfg->seconds;
  const int check_every = cfg->check_every > 0 ? cfg->check_every : 8;
  const bool fp8 = dtype == GMD_DT_FP8 || dtype == GMD_DT_MXFP8;
  const bool mx = dtype == GMD_DT_MXFP8;
  if (dtype != GMD_DT_BF16 && !fp8) return GMD_ERR_BADPARAM;
  if (m <= 0 || n <= 0 || k <= 0 || (m % (mx ? BM : 8)) || (n % (mx ? BN : 8)) || (k % (fp8 ? 128 : 64))) {
    gmd::set_error("GEMM shape: M and N in multiples of 8 (bf16, fp8) or 256 (MX-fp8), K in multiples of 64 (bf16) or 128 (fp8, MX-fp8)");
    return GMD_ERR_BADPARAM;
  }
  GMD_HIP(hipSetDevice(device));
  const size_t esz = fp8 ? 1 : 2;
  gmd::DeviceBuffer a, b, c, sums, sa_nat, sb_nat, sa_pk, sb_pk, refs;
  int rc;
  // Checksums land in a window of slots that is drained to the host (compared, cleared) whenever it is
  // full, so the soak runs for as long as it is asked to.
  const int kWindow = 1 << 13;
  const int kSamples = 4096;
  if ((rc = a.alloc((size_t)m * k * esz)) || (rc = b.alloc((size_t)n * k * esz)) || (rc = c.alloc((size_t)m * n * 2)) ||
      (rc = sums.alloc(sizeof(unsigned long long) * kWindow * kChecksumSlots * 8)) || (rc = refs.alloc(sizeof(float) * 2 * kSamples)))
    return rc;
  if (fp8) {
    fill_fp8_kernel<<<2048, 256>>>((uint8_t*)a.p, (uint64_t)m * k, 1);
    fill_fp8_kernel<<<2048, 256>>>((uint8_t*)b.p, (uint64_t)n * k, 2);
  } else {
    fill_bf16_kernel<<<2048, 256>>>((__bf16*)a.p, (uint64_t)m * k, 1);
    fill_bf16_kernel<<<2048, 256>>>((__bf16*)b.p, (uint64_t)n * k, 2);
  }
  if (mx) {  // one E8M0 scale per row and 32 k, row-major for the reference and packed for the kernel
    if ((rc = sa_nat.alloc((size_t)m * (k / 32))) || (rc = sb_nat.alloc((size_t)n * (k / 32))) || (rc = sa_pk.alloc((size_t)m * (k / 32))) ||
        (rc = sb_pk.alloc((size_t)n * (k / 32))))
      return rc;
    fill_e8m0_kernel<<<256, 256>>>((uint8_t*)sa_nat.p, (uint64_t)m * (k / 32), 3);
    fill_e8m0_kernel<<<256, 256>>>((uint8_t*)sb_nat.p, (uint64_t)n * (k / 32), 4);
    if ((rc = gmd_mx_pack_scales(sa_nat.p, sa_pk.p, m, k, nullptr)) || (rc = gmd_mx_pack_scales(sb_nat.p, sb_pk.p, n, k, nullptr))) return rc;
  }
  auto launch = [&]() { return gemm_launch_checked(dtype, a.p, b.p, sa_pk.p, sb_pk.p, c.p, m, n, k, nullptr); };
  GMD_HIP(hipMemset(sums.p, 0, sizeof(unsigned long long) * kWindow * kChecksumSlots * 8));
  GMD_HIP(hipGetLastError());
  if ((rc = launch())) return rc;  // warm-up; also the launch that is checked against the reference
  GMD_HIP(hipDeviceSynchronize());
  if (verify) {
    // kSamples entries of C against full-K dot products from a kernel that shares nothing with the GEMM.  bf16
    // rounding of C is 2^-9 relative; the tolerance is twice that, measured against |ref| + rms(ref) so that
    // entries near zero do not decide.
    gemm_reference_kernel<<<kSamples, 64>>>(dtype, a.p, b.p, (const uint8_t*)sa_nat.p, (const uint8_t*)sb_nat.p, (const __bf16*)c.p, m, n, k,
                                            0x5eed, (float*)refs.p, (float*)refs.p + kSamples);
    GMD_HIP(hipGetLastError());
    std::vector<float> h(2 * kSamples);
    GMD_HIP(hipMemcpy(h.data(), refs.p, h.size() * sizeof(float), hipMemcpyDeviceToHost));
    double sq = 0;
    for (int i = 0; i < kSamples; i++) sq += (double)h[i] * h[i];
    const double rms = std::sqrt(sq / kSamples);
    double worst = 0;
    bool finite = rms > 0 && std::isfinite(rms);
    for (int i = 0; i < kSamples && finite; i++) {
      if (!std::isfinite(h[kSamples + i])) finite = false;
      worst = std::max(worst, std::fabs((double)h[kSamples + i] - h[i]) / (std::fabs((double)h[i]) + rms));
    }
    out->max_rel_err = finite ? worst : INFINITY;
    if (!finite || worst > 4e-3) {
      out->verified = -1;
      gmd::set_error("GEMM stress: the first launch differs from the fp32 reference (max relative error %.3g over %d sampled entries)",
                     out->max_rel_err, kSamples);
      return GMD_ERR_VERIFY;
    }
  }

  // The soak proper.  The matrix pipes are to stay saturated for the whole run: GEMMs go back to back into one
  // stream, the host stays three batches ahead of the device and never drains the queue except to empty the
  // checksum window.  The checksum pass reads all of C (2 bytes per output element) and cannot share the compute
  // units with a GEMM whose waves hold the whole register file, so it sits in the same stream, after every
  // `check_every`-th GEMM: 31 us at 8192 x 8192 against 670 us per GEMM.
  const uint64_t n16 = (uint64_t)m * n * 2 / 16;
  constexpr int kMarkRing = 4;
  struct Rig {
    hipStream_t gemm = nullptr;
    hipEvent_t mark[kMarkRing + 1] = {};
    ~Rig() {
      for (hipEvent_t e : mark) if (e) (void)hipEventDestroy(e);
      if (gemm) (void)hipStreamDestroy(gemm);
    }
  } rig;
  GMD_HIP(hipStreamCreateWithFlags(&rig.gemm, hipStreamNonBlocking));
  for (auto& e : rig.mark) GMD_HIP(hipEventCreate(&e));
  // The register sum: every launch's output summed in the kernel's epilogue, harvested once per batch into pinned
  // host memory (slot b % ring, read after batch b's mark), held against batch x (the sum of one launch).
  // One sum per device and process: a second soak on the same card at the same time (two host threads on one GPU)
  // goes without and relies on the pass over C.
  static std::mutex soak_mu;
  static int soaks_on_device[64];
  struct Claim {
    int dev;
    bool first;
    explicit Claim(int d) : dev(d & 63) {
      std::lock_guard<std::mutex> l(soak_mu);
      first = soaks_on_device[dev]++ == 0;
    }
    ~Claim() {
      std::lock_guard<std::mutex> l(soak_mu);
      soaks_on_device[dev]--;
    }
  } claim(device);
  const bool reg = verify && claim.first && reg_sum_shape(dtype, m, n, k) && !getenv("GPUMON_GEMM_NO_REGSUM");
  struct Pinned {
    unsigned long long* p = nullptr;
    ~Pinned() { if (p) (void)hipHostFree(p); }
  } harvest;
  unsigned long long* harvest_dev = nullptr;
  unsigned long long reg_ref = 0;
  if (reg) {
    GMD_HIP(hipHostMalloc((void**)&harvest.p, (kMarkRing + 2) * sizeof(unsigned long long), hipHostMallocDefault));
    GMD_HIP(hipHostGetDevicePointer((void**)&harvest_dev, harvest.p, 0));
    reg_sum_harvest_kernel<<<1, kRegSumSlots, 0, rig.gemm>>>(harvest_dev + kMarkRing + 1);  // start from zero
    if ((rc = gemm_launch_checked(dtype, a.p, b.p, sa_pk.p, sb_pk.p, c.p, m, n, k, rig.gemm, true))) return rc;
    reg_sum_harvest_kernel<<<1, kRegSumSlots, 0, rig.gemm>>>(harvest_dev + kMarkRing + 1);
    GMD_HIP(hipGetLastError());
    GMD_HIP(hipStreamSynchronize(rig.gemm));
    reg_ref = harvest.p[kMarkRing + 1];
    if (!reg_ref) {
      gmd::set_error("GEMM stress: the kernel's own output sum is zero");
      return GMD_ERR_VERIFY;
    }
  }
  const double flops = 2.0 * m * n * (double)k;
  const double t_start = now_s();
  double gpu_s = 0, best_ms = 1e30;
  uint64_t launches = 0, in_window = 0, batches = 0;
  unsigned long long ref_sum = 0;
  bool have_ref = false;
  constexpr size_t kSumWords = kChecksumSlots * 8;  // words per checksum in `sums`
  std::vector<unsigned long long> h((size_t)kWindow * kSumWords);
  auto drain = [&]() -> int {  // both streams idle: compare the window against the first checksum, clear it
    if (!verify || !in_window) return GMD_OK;
    GMD_HIP(hipStreamSynchronize(rig.gemm));
    GMD_HIP(hipMemcpy(h.data(), sums.p, in_window * kSumWords * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (uint64_t i = 0; i < in_window; i++) {
      unsigned long long sum = 0;
      for (int slot = 0; slot < kChecksumSlots; slot++) sum += h[i * kSumWords + slot * 8];
      if (!have_ref) ref_sum = sum, have_ref = true;
      if (sum != ref_sum) out->sdc_mismatches++;
    }
    GMD_HIP(hipMemset(sums.p, 0, in_window * kSumWords * sizeof(unsigned long long)));
    in_window = 0;
    return GMD_OK;
  };
  // Launches per batch (one harvest of the running sum, one mark): eight of the large ones; a small GEMM's launches are
  // over in tens of microseconds, and the harvest kernel's four should stay a fraction of a percent of a batch.
  const int batch = flops < 2e11 ? 32 : 8;
  // mark[b % (kRing + 1)] closes batch b; batch b's time is the distance between marks b-1 and b.
  auto settle = [&](uint64_t b) -> int {
    hipEvent_t e1 = rig.mark[b % (kMarkRing + 1)], e0 = rig.mark[(b + kMarkRing) % (kMarkRing + 1)];
    GMD_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GMD_HIP(hipEventElapsedTime(&ms, e0, e1));
    gpu_s += ms * 1e-3;
    best_ms = std::min(best_ms, (double)ms / batch);
    if (reg) {
      if (harvest.p[b % (kMarkRing + 1)] != (unsigned long long)batch * reg_ref) out->sdc_mismatches++;
      out->checked_in_kernel += batch;
    }
    return GMD_OK;
  };
  // $GPUMON_GEMM_INJECT_AT=N (tests): the exponent's top bit of A's first element is flipped for launch N alone, so
  // that exactly one launch of the soak computes something else.
  const char* inject_env = getenv("GPUMON_GEMM_INJECT_AT");
  const long long inject_at = inject_env ? atoll(inject_env) : -1;
  const unsigned inject_mask = dtype == GMD_DT_BF16 ? 0x4000u : 0x40u;
  GMD_HIP(hipEventRecord(rig.mark[0], rig.gemm));
  uint64_t settled = 0;  // batches 1..settled are accounted for
  for (;;) {
    for (int i = 0; i < batch; i++, launches++) {
      if (inject_at >= 0 && ((long long)launches == inject_at || (long long)launches == inject_at + 1))
        flip_bits_kernel<<<1, 1, 0, rig.gemm>>>((unsigned*)a.p, inject_mask);
      if ((rc = gemm_launch_checked(dtype, a.p, b.p, sa_pk.p, sb_pk.p, c.p, m, n, k, rig.gemm, reg))) return rc;
      if (verify && launches % check_every == 0) {
        checksum_kernel<<<checksum_grid(n16), 256, 0, rig.gemm>>>((const uint4*)c.p, n16, (unsigned long long*)sums.p + in_window * kSumWords);
        in_window++;
        out->checked_in_memory++;
      }
    }
    batches++;
    if (reg) reg_sum_harvest_kernel<<<1, kRegSumSlots, 0, rig.gemm>>>(harvest_dev + batches % (kMarkRing + 1));
    GMD_HIP(hipEventRecord(rig.mark[batches % (kMarkRing + 1)], rig.gemm));
    // at most kRing - 1 batches in flight: the slot that the next mark takes has been read by then
    while (batches - settled >= (uint64_t)kMarkRing - 1) {
      if ((rc = settle(settled + 1))) return rc;
      settled++;
    }
    if (in_window + batch > (uint64_t)kWindow) {
      // Emptying the window idles the device once per kWindow checksums; the idle time is not GEMM time.
      while (settled < batches) {
        if ((rc = settle(settled + 1))) return rc;
        settled++;
      }
      if ((rc = drain())) return rc;
      GMD_HIP(hipEventRecord(rig.mark[batches % (kMarkRing + 1)], rig.gemm));
    }
    if (now_s() - t_start >= seconds) break;
  }
  while (settled < batches) {
    if ((rc = settle(settled + 1))) return rc;
    settled++;
  }
  GMD_HIP(hipStreamSynchronize(rig.gemm));
  GMD_HIP(hipGetLastError());
  out->launches = launches;
  out->seconds = gpu_s;
  out->tflops = flops * (double)launches / gpu_s / 1e12;
  out->tflops_best = flops / (best_ms * 1e-3) / 1e12;
  if (verify) {
    if ((rc = drain())) return rc;
    out->verified = out->sdc_mismatches ? -1 : 1;
    if (ref_sum == 0) out->verified = -1;  // an all-zero C means the kernel did nothing
  }
  if (out->verified < 0) {
    gmd::set_error("GEMM stress: %llu of %llu launches differ from the first (silent data corruption?)",
                   (unsigned long long)out->sdc_mismatches, (unsigned long long)launches);
    return GMD_ERR_VERIFY;
  }
  return GMD_OK;
}

int gmd_mfma_peak(int device, int dtype, double seconds, gmd_mfma_result_t* out) {
  if (!out || dtype < 0 || dtype > GMD_DT_MXFP8) return GMD_ERR_BADPARAM;
  memset(out, 0, sizeof *out);
  GMD_HIP(hipSetDevice(device));
  const int blocks = gmd::kCUs * 2, threads = 256, iters = 4096;  // 2 waves per SIMD
  gmd::DeviceBuffer sink, stamps;
  int rc;
  if ((rc = sink.alloc(sizeof(float) * blocks * threads)) || (rc = stamps.alloc(sizeof(unsigned long long) * 2 * blocks)))
    return rc;
  auto run = [&](unsigned long long* st) {
    switch (dtype) {
      case GMD_DT_BF16: peak_kernel<GMD_DT_BF16><<<blocks, threads>>>((float*)sink.p, iters, st); break;
      case GMD_DT_FP8: peak_kernel<GMD_DT_FP8><<<blocks, threads>>>((float*)sink.p, iters, st); break;
      case GMD_DT_MXFP8: peak_kernel<GMD_DT_MXFP8><<<blocks, threads>>>((float*)sink.p, iters, st); break;
      default: peak_kernel<GMD_DT_F32><<<blocks, threads>>>((float*)sink.p, iters, st); break;
    }
  };
  run(nullptr);
  GMD_HIP(hipGetLastError());
  GMD_HIP(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  GMD_HIP(hipEventCreate(&e0));
  GMD_HIP(hipEventCreate(&e1));
  const double flops_launch = (double)blocks * (threads / 64) * (double)iters * 8 * peak_flops_per_mfma(dtype);
  double t_start = now_s(), gpu_s = 0, best_ms = 1e30;
  uint64_t launches = 0;
  do {
    GMD_HIP(hipEventRecord(e0, 0));
    for (int i = 0; i < 4; i++) run(nullptr);
    GMD_HIP(hipEventRecord(e1, 0));
    GMD_HIP(hipEventSynchronize(e1));
    float ms = 0;
    GMD_HIP(hipEventElapsedTime(&ms, e0, e1));
    gpu_s += ms * 1e-3;
    best_ms = std::min(best_ms, (double)ms / 4);
    launches += 4;
  } while (now_s() - t_start < seconds);
  // One stamped launch after the soak: shader cycles per 100 MHz reference tick.
  run((unsigned long long*)stamps.p);
  GMD_HIP(hipDeviceSynchronize());
  std::vector<unsigned long long> h(2 * blocks);
  GMD_HIP(hipMemcpy(h.data(), stamps.p, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::vector<double> mhz;
  for (int i = 0; i < blocks; i++)
    if (h[2 * i + 1]) mhz.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0);
  if (!mhz.empty()) {
    std::sort(mhz.begin(), mhz.end());
    out->clock_mhz_est = mhz[mhz.size() / 2];
  }
  GMD_HIP(hipGetLastError());
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  out->launches = launches;
  out->seconds = gpu_s;
  out->tflops = flops_launch * (double)launches / gpu_s / 1e12;
  out->tflops_best = flops_launch / (best_ms * 1e-3) / 1e12;
  return GMD_OK;
}

}  // extern "C"
