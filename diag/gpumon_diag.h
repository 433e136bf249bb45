/*
 * gpumon_diag.h — C ABI of the active diagnostics (libgpumon_diag.so).
 *
 * Hand-written CDNA4 (gfx950) HIP kernels in the role of DCGM's diagnostic
 * plugins, which the reference depends on being installed but never wraps:
 *   - HBM stream bandwidth   (copy / scale / add / triad / read / write)
 *   - memtest pattern sweep  (write + verify over most of the 288 GB HBM3E)
 *   - MFMA stress            (register-resident matrix-pipe peak, and a
 *                             LDS-tiled bf16 / fp8 GEMM with an SDC checksum)
 * Every entry point is synchronous, selects `device` itself and returns 0 or a
 * negative gmd error; pointer-taking variants run on caller-owned device
 * memory so tests can check numerics against an independent reference.
 */
#ifndef GPUMON_DIAG_H_
#define GPUMON_DIAG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GMD_API __attribute__((visibility("default")))

enum {
  GMD_OK = 0,
  GMD_ERR_BADPARAM = -1,
  GMD_ERR_HIP = -2,      /* a HIP call failed; see gmd_last_error() */
  GMD_ERR_NO_DEVICE = -3,
  GMD_ERR_NOMEM = -4,
  GMD_ERR_VERIFY = -5,   /* the kernel ran but its result is wrong */
  GMD_ERR_UNSUPPORTED = -6 /* a run-time dependency (librccl) is missing */
};

GMD_API const char* gmd_last_error(void);
GMD_API int gmd_device_count(void);
typedef struct {
  char name[128];
  char gcn_arch[64];
  uint64_t total_mem, free_mem;
  int compute_units, clock_mhz, mem_clock_mhz, l2_bytes, lds_per_cu;
  char pci_bdf[32];
} gmd_device_info_t;
GMD_API int gmd_device_info(int device, gmd_device_info_t* out);

/* ------------------------------------------------------------ HBM stream -- */
typedef enum {
  GMD_STREAM_COPY = 0,  /* c = a            2 x bytes */
  GMD_STREAM_SCALE = 1, /* b = s * c        2 x bytes */
  GMD_STREAM_ADD = 2,   /* c = a + b        3 x bytes */
  GMD_STREAM_TRIAD = 3, /* a = b + s * c    3 x bytes */
  GMD_STREAM_READ = 4,  /* sum(a)           1 x bytes */
  GMD_STREAM_WRITE = 5, /* a = s            1 x bytes */
  GMD_STREAM_KERNELS = 6
} gmd_stream_kernel_t;

typedef struct {
  double tbps_best;     /* 1e12 bytes per second, best iteration */
  double tbps_median;
  double ms_best;
  uint64_t bytes_per_iter; /* bytes read + written by one launch */
  int32_t verified;        /* 1 = sampled results checked and correct */
  int32_t iters;
} gmd_stream_result_t;

/* Allocates its own buffers of `bytes_per_buffer` each (use >= 1 GiB: the
 * 256 MiB Infinity Cache must not hold the working set). */
GMD_API int gmd_hbm_stream(int device, int kernel, uint64_t bytes_per_buffer, int iters, int nontemporal,
                           gmd_stream_result_t* out);
/* One launch on caller buffers (float elements, 16-byte aligned, count a multiple of 4).
 * READ: `c`, when not NULL, is a sink of GMD_STREAM_SINK_FLOATS floats into which every
 * workgroup ADDS its sum (slot = workgroup index modulo the slot count); zero it beforehand if
 * the total matters.  The launch is then fully asynchronous. */
#define GMD_STREAM_SINK_FLOATS 32768
GMD_API int gmd_hbm_stream_launch(int kernel, void* a, void* b, void* c, float scalar, uint64_t n_floats,
                                  int nontemporal, void* hip_stream);

/* --------------------------------------------------------------- memtest -- */
#define GMD_PAT_WALKING_ONES (1u << 0)
#define GMD_PAT_WALKING_ZEROS (1u << 1)
#define GMD_PAT_ADDRESS (1u << 2)      /* every word holds its own byte offset in the region */
#define GMD_PAT_CHECKERBOARD (1u << 3) /* 0xAAAA... */
#define GMD_PAT_INV_CHECKERBOARD (1u << 4)
#define GMD_PAT_RANDOM (1u << 5)       /* counter-based hash of (index, seed) */
#define GMD_PAT_MODULO_20 (1u << 6)    /* one word in 20 differs */
#define GMD_PAT_ALL 0x7fu
#define GMD_MEMTEST_LOG 32

typedef struct {
  uint64_t byte_offset; /* offset inside the tested region */
  uint64_t expected, actual;
  uint32_t pattern; /* one GMD_PAT_* bit */
  uint32_t chunk;   /* which allocation of the region */
  uint64_t device_address; /* virtual address of the word in this process */
  uint32_t flipped_bits;   /* bits that differ */
  uint32_t page_offset;    /* device_address modulo 4096, the granule in which consecutive addresses change HBM channel */
} gmd_memtest_error_t;

typedef struct {
  uint64_t bytes_tested;   /* size of the region under test */
  uint64_t bytes_requested;
  uint64_t errors;         /* mismatching 64-bit words, all patterns */
  uint32_t patterns_run;
  uint32_t chunks;
  double write_tbps, verify_tbps; /* mean over patterns */
  double seconds;
  uint32_t logged;
  gmd_memtest_error_t log[GMD_MEMTEST_LOG];
  uint32_t scrubbed;     /* 1 = twice the Infinity Cache was overwritten between every write and its check, so that what
                            was checked came from the memory cells; 0 = no memory left for the scrub buffer */
  uint32_t march_passes; /* march sweeps completed */
  double march_tbps;     /* bytes read + written per second by the march elements */
} gmd_memtest_result_t;

typedef void (*gmd_progress_cb)(double fraction, const char* what, void* user);

typedef struct {
  uint64_t bytes;        /* region size; 0 = fraction of the free HBM */
  double fraction;       /* used when bytes == 0; default 0.90 */
  uint64_t chunk_bytes;  /* allocation granule; default 2 GiB */
  uint32_t patterns;     /* mask of GMD_PAT_*; 0 = all */
  uint32_t passes;       /* repeats of the whole sweep; default 1 */
  uint64_t seed;
  /* Fault injection for tests: XOR `inject_xor` into the 64-bit word at
   * `inject_offset` between the write and the verify pass of every pattern.
   * inject_xor == 0 disables it. */
  uint64_t inject_offset;
  uint64_t inject_xor;
  gmd_progress_cb progress;
  void* progress_user;
  /* 1 = after the patterns, a march over the region: write, then read-check-write-complement going up, the same
   * going down, and a last check (each element one kernel: a read+write stress as well). */
  uint32_t march;
} gmd_memtest_config_t;

GMD_API int gmd_memtest(int device, const gmd_memtest_config_t* cfg, gmd_memtest_result_t* out);
/* Single passes on a caller buffer (n_words 64-bit words, 16-byte aligned, even count). */
GMD_API int gmd_memtest_fill(void* buf, uint64_t n_words, uint64_t base_word, uint32_t pattern, uint64_t seed,
                             void* hip_stream);
GMD_API int gmd_memtest_verify(const void* buf, uint64_t n_words, uint64_t base_word, uint32_t pattern, uint64_t seed,
                               uint64_t* errors_out, gmd_memtest_error_t* log, uint32_t* logged, void* hip_stream);

/* ----------------------------------------------------------- MFMA stress -- */
typedef enum { GMD_DT_BF16 = 0, GMD_DT_FP8 = 1, GMD_DT_F32 = 2, GMD_DT_MXFP8 = 3 } gmd_dtype_t;

typedef struct {
  double tflops;       /* dense, 2*M*N*K per GEMM (or MFMA flops for the pipe test) */
  double tflops_best;  /* best single launch */
  double seconds;
  uint64_t launches;
  double clock_mhz_est; /* shader clock held during the run, from s_memtime / s_memrealtime; 0 = not measured */
  int32_t verified;     /* 1 = checksum / reference comparison passed, 0 = not checked, -1 = failed */
  uint64_t sdc_mismatches; /* launches whose checksum differed from the first launch */
  double max_rel_err;
  uint64_t checked_in_kernel;  /* launches whose output was summed in the kernel's epilogue and held against the first */
  uint64_t checked_in_memory;  /* launches followed by the checksum pass over C in memory */
} gmd_mfma_result_t;

/* Matrix-pipe peak: MFMAs on register-resident operands, no memory traffic. */
GMD_API int gmd_mfma_peak(int device, int dtype, double seconds, gmd_mfma_result_t* out);
/* GEMM stress: C[M,N] = A[M,K] * B[N,K]^T on random data for `seconds` (>= 1 launch).  With `verify`, the
 * first launch is compared with an independent reference (4096 sampled entries of C against full-K dot
 * products computed by a separate kernel in double precision: max_rel_err, tolerance 4e-3 of |ref| + rms) and
 * every later launch's output checksum with the first one's (silent-data-corruption check).  For whole-tile shapes
 * the soak's kernel sums the words it stores on their way out (one atomic per wave), so EVERY launch is held against
 * the first at no cost in throughput; the pass over C in memory, which also sees the stores and the memory cells,
 * follows every `check_every`-th launch. */
GMD_API int gmd_gemm_stress(int device, int dtype, int m, int n, int k, double seconds, int verify,
                            gmd_mfma_result_t* out);
/* The same with every knob.  GEMMs go back to back into one stream with the host three batches ahead; the checksum
 * pass over C follows every `check_every`-th GEMM in the same stream (it cannot overlap a GEMM whose waves hold the
 * whole register file; at 8192^3 it takes 4.6 % of a GEMM's time, so one in four keeps the pipes 98.9 % on GEMMs). */
typedef struct {
  int32_t device, dtype, m, n, k;
  int32_t verify;       /* reference comparison of the first launch + checksums */
  int32_t check_every;  /* checksum every Nth launch; <= 0: 8 */
  double seconds;
} gmd_gemm_stress_config_t;
GMD_API int gmd_gemm_stress_run(const gmd_gemm_stress_config_t* cfg, gmd_mfma_result_t* out);
/* One GEMM on caller memory.  A: [M,K] row-major, B: [N,K] row-major (i.e. B^T), C: [M,N] row-major bf16.
 * GMD_DT_BF16: A, B bf16.  GMD_DT_FP8: A, B OCP e4m3, unscaled (mfma_f32_16x16x128_f8f6f4, twice the rate of
 * mfma_f32_16x16x32_fp8_fp8, which schedules 0, 1, 3, 4, 5 and 7 run).
 * M, N multiples of 8 (tiles that reach beyond the matrix and odd K-tile counts run a general instantiation of the
 * four-wave kernel); K a multiple of 64 (bf16) / 128 (fp8), i.e. whole 128-byte K-tiles.  GMD_DT_MXFP8 needs scales
 * and M, N in multiples of 256: gmd_gemm_launch_mx. */
GMD_API int gmd_gemm_launch(int dtype, const void* a, const void* b, void* c, int m, int n, int k, void* hip_stream);
/* OCP MX-fp8 GEMM: e4m3 elements with one E8M0 scale (2^(e-127)) per row and 32 consecutive k, on the
 * block-scaled MFMA (mfma_scale_f32_16x16x128_f8f6f4).  The scale tensors are in the packed form that
 * gmd_mx_pack_scales produces from row-major uint8 [rows][K/32] (rows a multiple of 128, K of 128; the packed
 * tensor has the same number of bytes and must be 4-byte aligned). */
GMD_API int gmd_gemm_launch_mx(const void* a, const void* b, const void* scale_a_packed, const void* scale_b_packed, void* c,
                               int m, int n, int k, void* hip_stream);
GMD_API int gmd_mx_pack_scales(const void* scales_row_major, void* packed, int rows, int k, void* hip_stream);

/* GEMM schedule: -1 = the fastest for the dtype and shape (default); 0 = simple double-buffered loop;
 * 1 = eight waves x 128x64, one workgroup per tile; 2 = four waves x 128x128 with the accumulators in AGPRs;
 * 3 = eight waves x 128x64 as one stream per CU over its tiles (needs more tiles than CUs and >= 4 K-tiles);
 * 4 / 5 = schedule 1 with the narrow / the cached epilogue (A/B references); 6 = schedule 2 as one stream per
 * CU (A/B reference, same conditions as 3, otherwise 2 runs).  Partial tiles run the general four-wave kernel
 * whatever the schedule, an odd number of K-tiles does unless the schedule is 0.  MX-fp8 runs 2, 6 or 0.
 * 7 = schedule 2 on the K=32 instruction for unscaled fp8 (2 for the other data types).
 * 8 = 128 x 128 tiles, four waves of 64 x 64, two workgroups per CU (bf16 and fp8, any M, N in multiples of 8): what
 * the default picks where the 256 x 256 tiles cannot fill the chip (2048^3 is 64 of them on 256 CUs); MX-fp8 runs 2.
 * 9 = schedule 8 split along K: several workgroups share a tile, each over a range of K-tiles, and a second kernel adds
 * their fp32 sums up in a fixed order; 10 = the same on the 256 x 256 tiles of schedule 2 (whole tiles, K-tiles in pairs;
 * MX-fp8 as well, which has no schedule 9).
 * Asked for, they split into as many ranges as fit (at most 8 and one workgroup per CU), or into gmd_gemm_set_split()
 * ranges; where a shape cannot be split, 8 and 2 run.  The default takes one of them where a model of the launch times
 * has it ahead (few tiles and a long K: 1024 x 1024 x 4096, 2048 x 2048 x 8192); gmd_gemm_plan() says what a shape gets.
 * All schedules but 9 and 10 give
 * bit-identical results, except that unscaled fp8 has two groups of them, one per MFMA instruction (K=128: -1, 2, 6, 8;
 * K=32: the others), whose sums differ in the order of their additions; a split GEMM adds the same products in
 * another order again, the same one at every launch.  Also settable with the environment variable GMD_GEMM_VARIANT before the first launch. */
GMD_API int gmd_gemm_set_variant(int variant);
/* What the schedule in force does with a GEMM of this shape on the current device: *schedule = 9 or 10 and *splits = the
 * number of K ranges, or 0 and 1 (not split).  gmd_gemm_split_k returns the number of ranges alone. */
GMD_API int gmd_gemm_plan(int dtype, int m, int n, int k, int* schedule, int* splits);
/* Number of K ranges schedules 9 and 10 use when asked for: 2..8, or 0 for as many as fit (the default). */
GMD_API int gmd_gemm_set_split(int splits);
GMD_API int gmd_gemm_split_k(int dtype, int m, int n, int k);

/* Timing-only ablation builds (make diag-ablate): per-workgroup pairs (shader-clock ticks,
 * 100 MHz real-time ticks) of the last launch of a stamped variant.  GMD_ERR_UNSUPPORTED in
 * the product library. */
GMD_API int gmd_gemm_read_stamps(unsigned long long* out, int n_pairs);
GMD_API int gmd_gemm_read_timeline(unsigned long long* out, int n_tiles);

/* ---------------------------------------------------------- interconnect -- */
/* Single-process RCCL sweep: one communicator per listed GPU (ncclCommInitAll),
 * collectives issued as one group per iteration.  librccl is loaded at run time.
 * payload bytes: all_reduce / broadcast = the buffer; all_gather / reduce_scatter =
 * the gathered (total) buffer; all_to_all = one GPU's send buffer. */
typedef enum {
  GMD_COLL_ALL_REDUCE = 0,
  GMD_COLL_ALL_GATHER = 1,
  GMD_COLL_REDUCE_SCATTER = 2,
  GMD_COLL_ALL_TO_ALL = 3,
  GMD_COLL_BROADCAST = 4,
  GMD_COLL_COUNT = 5
} gmd_collective_t;

typedef struct {
  int32_t ndev;          /* 0 = all visible GPUs */
  const int* devices;    /* NULL = 0..ndev-1 */
  uint32_t ops_mask;     /* bit per gmd_collective_t; 0 = all */
  uint64_t min_bytes;    /* default 8 */
  uint64_t max_bytes;    /* default 1 GiB */
  int32_t factor;        /* size step, default 4 */
  int32_t iters;         /* 0 = size dependent (50 / 20 / 8) */
  int32_t warmup;        /* default 3 */
  int32_t verify;        /* 1 = check every size once on patterned data */
} gmd_rccl_config_t;

typedef struct {
  int32_t op;
  int32_t ndev;
  uint64_t bytes;
  int32_t iters;
  int32_t verified;      /* 1 ok, 0 not checked, -1 mismatch */
  double time_us;
  double algbw_gbps;     /* 1e9 bytes/s */
  double busbw_gbps;
} gmd_rccl_point_t;

GMD_API const char* gmd_collective_name(int op);
GMD_API double gmd_busbw_factor(int op, int ndev);
/* *n: capacity of out on entry, points written on return. */
GMD_API int gmd_rccl_sweep(const gmd_rccl_config_t* cfg, gmd_rccl_point_t* out, int* n);

/* Link test: the compute units of one GPU (the initiator) copy between their own memory and a peer's, 16 bytes
 * per lane with streaming accesses, so the figure is the link's and not the copy engines'.  Push (default): local
 * loads, stores into the peer's memory.  Pull: loads from the peer's memory, local stores (what all-gather-by-read
 * and P2P reads do).  Three traffic patterns over the same buffers:
 *   UNI   one ordered pair at a time                 -> what one direction of one link carries
 *   BIDI  both directions of a pair at once          -> whether the link holds its rate full duplex
 *   ALL   every GPU works on every peer at once      -> whether the links of a GPU are independent
 * Every GPU runs ONE launch per pattern step that serves all of its peers of that step (workgroup sets per peer), so
 * ALL keeps seven links busy from one hardware queue; each pair is timed inside that launch by the earliest entry
 * and latest exit of its own workgroups on the GPU's constant-rate real-time counter.
 * gbps[i*ndev+j] is the rate at which devices[i] wrote to (push) or read from (pull) devices[j] under that pattern
 * (1e9 bytes/s), 0 on the diagonal, -1 without peer access.  With `verify`, bad[i*ndev+j] counts the 16-byte words
 * that arrived different from what was sent (bad may be NULL).  One GPU: the 1x1 matrix {0}, GMD_OK. */
#define GMD_LINK_MAX_PEERS 63
typedef enum { GMD_LINK_UNI = 0, GMD_LINK_BIDI = 1, GMD_LINK_ALL = 2 } gmd_link_mode_t;
typedef struct {
  int32_t ndev;        /* 0 = all visible GPUs */
  const int* devices;  /* NULL = 0..ndev-1 */
  uint64_t bytes;      /* per transfer; default 256 MiB */
  int32_t iters;       /* timed passes over the buffer per pair; default 5 */
  int32_t verify;      /* 1 = check the data at the destination after the timed transfers */
  int32_t workgroups;  /* workgroups per peer (256 lanes each); default 64 */
  int32_t pull;        /* 1 = remote loads, local stores */
  double* per_gpu_gbps; /* ALL only, may be NULL: ndev entries, bytes the GPU's one launch moved over the time between
                           the HIP events around it (all peers together; an independent clock for the stamps) */
} gmd_link_config_t;
GMD_API int gmd_link_test(const gmd_link_config_t* cfg, int mode, double* gbps, uint64_t* bad);

/* The link test's copy and check kernels run inside one GPU (local memory to local memory): proves the kernels on a
 * host where no peer is visible.  bytes need not be a multiple of the kernel's stride; *bad counts wrong words and a
 * touched guard word past the end. */
GMD_API int gmd_link_selftest(int device, uint64_t bytes, int workgroups, double* gbps, uint64_t* bad);

/* The fused all-peers launch inside one GPU: `npeers` local sources and destinations stand in for the peers.  gbps and
 * bad take npeers entries.  The result relates the two clocks: every peer's stamp window must lie inside the time
 * between the HIP events around the launch. */
typedef struct {
  int32_t npeers;
  double launch_ms;                    /* HIP events around the timed launch */
  double window_ms_min, window_ms_max; /* shortest and longest per-peer stamp window */
  double launch_gbps;                  /* all peers' bytes over launch_ms */
  double tick_hz;                      /* rate of the real-time counter the stamps are in */
} gmd_link_fan_result_t;
GMD_API int gmd_link_fan_selftest(int device, int npeers, uint64_t bytes, int workgroups, int iters, double* gbps, uint64_t* bad,
                                  gmd_link_fan_result_t* result);

/* Grading of one link matrix, host arithmetic only (no GPU needed).  A pair fails when it has no peer access,
 * delivered corrupt data, runs below the absolute floor, or below fail_ratio of the median of the other pairs;
 * it warns below warn_ratio of that median.  The sibling median needs at least two other measured pairs;
 * with fewer only the floor applies.  *n: capacity of out on entry, verdicts written on return (one per ordered
 * pair, ndev*(ndev-1); none for one GPU).  Returns the worst health. */
typedef struct {
  double floor_gbps;  /* <= 0: no absolute floor */
  double warn_ratio;  /* <= 0: 0.90 */
  double fail_ratio;  /* <= 0: 0.75 */
} gmd_link_rule_t;
typedef struct {
  int32_t src, dst;          /* positions in the device list */
  int32_t health;            /* 0 pass, 10 warn, 20 fail (the engine's health scale) */
  double gbps;
  double sibling_median_gbps; /* 0 when there were not enough siblings */
  char message[160];
} gmd_link_verdict_t;
GMD_API int gmd_link_grade(int ndev, const double* gbps, const uint64_t* bad, const gmd_link_rule_t* rule,
                           gmd_link_verdict_t* out, int* n);

/* What a healthy node is expected to deliver, derived from the advertised rate of one direction of one link (the
 * engine's link table has it; 76.8 GB/s on MI355X, whose 153.6 GB/s per link counts both directions).  The
 * deratings say which fraction of that rate a copy kernel is expected to reach per direction under each traffic
 * pattern; a measurement warns below warn_ratio and fails below fail_ratio of its expectation.  Members <= 0 take
 * the defaults given here. */
typedef struct {
  double link_gbps;         /* one direction of one link, 1e9 bytes/s; default 76.8 */
  double derate_uni;        /* one pair at a time; default 0.70 */
  double derate_bidi;       /* per direction with both busy; default 0.65 */
  double derate_all;        /* per link with every link of every GPU busy; default 0.60 */
  double pull_factor;       /* remote loads against remote stores; default 0.80 */
  double derate_collective; /* collectives that can use all n-1 links of a GPU, per link; default 0.50 */
  double warn_ratio;        /* default 0.90 */
  double fail_ratio;        /* default 0.75 */
  double scale;             /* multiplies every expectation (a stricter or laxer site, tests); default 1 */
  int32_t pull;             /* the matrices were measured with remote loads */
} gmd_link_expect_t;
GMD_API void gmd_link_expect_defaults(gmd_link_expect_t* e);
/* Expected rate per direction of one link under a pattern, times `peers` (>= 1). */
GMD_API double gmd_link_expected_gbps(const gmd_link_expect_t* e, int mode, int peers);
/* Expected peak bus bandwidth of a collective on ndev GPUs (0 for one GPU: nothing crosses a link).  All-reduce
 * and broadcast: one direction of one link at the one-pair derating (a ring is bound by one link per hop, and
 * nothing the library builds on the mesh may be slower than that).  All-gather, reduce-scatter, all-to-all: the
 * n-1 links of a GPU together at the collective derating. */
GMD_API double gmd_rccl_expected_busbw(const gmd_link_expect_t* e, int op, int ndev);
/* 0 / 10 / 20 for a value against its expectation (0 when there is no expectation). */
GMD_API int gmd_grade_against(double value, double expected, double warn_ratio, double fail_ratio);
/* Each value against the median of the others (GPUs of one node under the same load): warn below warn_ratio
 * (default 0.95), fail below fail_ratio (default 0.90).  health and others_median (may be NULL) take n entries;
 * one value alone passes.  Returns the worst health. */
GMD_API int gmd_outlier_grade(int n, const double* values, double warn_ratio, double fail_ratio, int32_t* health,
                              double* others_median);

/* The three matrices of a link test read together (host arithmetic only).  A finding names what is wrong and where:
 *   no_access, corrupt_data, slow_link (one pair at a time: against the expectation and against the other links),
 *   half_duplex (a direction that loses its rate when the other is busy), low_egress / low_ingress (a GPU that
 *   moves less than the others, or than expected, as initiator / as peer with every link busy; dst / src = -1),
 *   slow_under_load (a pair that falls behind its GPU's other links only with every link busy).
 * A pair with a finding is left out of the busier patterns' statistics, so one fault gives one finding.  bidi and
 * all may be NULL (pattern not run); bad counts corrupt words per pair over all patterns (may be NULL).
 * *n: capacity of out on entry, findings written on return.  Returns the worst health, or an error. */
typedef enum {
  GMD_LINK_NO_ACCESS = 0,
  GMD_LINK_CORRUPT = 1,
  GMD_LINK_SLOW = 2,
  GMD_LINK_HALF_DUPLEX = 3,
  GMD_LINK_SLOW_UNDER_LOAD = 4,
  GMD_LINK_LOW_EGRESS = 5,
  GMD_LINK_LOW_INGRESS = 6,
  GMD_LINK_FINDING_KINDS = 7
} gmd_link_finding_kind_t;
typedef struct {
  int32_t kind, mode;  /* gmd_link_finding_kind_t, gmd_link_mode_t */
  int32_t src, dst;    /* initiator and peer positions in the device list; -1 = the whole GPU on the other side */
  int32_t health;      /* 10 warn, 20 fail */
  double gbps, expected_gbps;
  char message[192];
} gmd_link_finding_t;
GMD_API const char* gmd_link_finding_name(int kind);
GMD_API int gmd_link_diagnose(int ndev, const double* uni, const double* bidi, const double* all, const uint64_t* bad,
                              const gmd_link_expect_t* expect, gmd_link_finding_t* out, int* n);

/* Pairwise copy bandwidth over the GPU fabric with peer access (hipMemcpyPeerAsync):
 * gbps[i*ndev+j] = devices[i] -> devices[j], 0 on the diagonal, -1 where peer access
 * is not available.  On MI355X each ordered pair is one xGMI link. */
GMD_API int gmd_p2p_matrix(int ndev, const int* devices, uint64_t bytes, int iters, double* gbps);

/* ------------------------------------------------------------- host link -- */
/* The PCIe path between pinned host memory and one GPU, loaded five ways (diag/hip/pcie_test.hip): copy engines
 * host-to-device, device-to-host and both at once; a kernel reading host memory and a kernel writing it.  Every
 * transfer carries a pattern and is checked on the far side.  Rates are GB/s (1e9 bytes) of the best iteration. */
typedef struct {
  uint64_t bytes;      /* per transfer; 0 = 256 MiB */
  int32_t iters;       /* <= 0: 5 */
  int32_t verify;      /* check the data on the far side */
  int32_t workgroups;  /* of the two kernels; <= 0: 1024 */
  int32_t reserved;
} gmd_pcie_config_t;
typedef struct {
  double h2d_gbps, d2h_gbps;              /* copy engines, one direction at a time */
  double bidir_h2d_gbps, bidir_d2h_gbps;  /* copy engines, both directions at once (whether they overlap is the runtime's choice of engines) */
  double kernel_read_gbps, kernel_write_gbps;
  double duplex_read_gbps, duplex_write_gbps;  /* the two kernels at once: the link's rate with both directions busy */
  double small_h2d_us, small_d2h_us;      /* an 8-byte blocking copy, host clock, best of 56 */
  uint64_t bad_h2d, bad_d2h, bad_kernel_read, bad_kernel_write;  /* 16-byte words that arrived wrong */
  uint64_t bytes;
  int32_t iters, verified;
} gmd_pcie_result_t;
GMD_API int gmd_pcie_test(int device, const gmd_pcie_config_t* cfg, gmd_pcie_result_t* out);
/* What one direction of a link of that generation and width carries at most, GB/s (the reference's per-lane table,
 * bindings/go/nvml/nvml.go:314-326, continued to generation 6); 0 for values outside the table. */
GMD_API double gmd_pcie_link_gbps(int gen, int width);

#ifdef __cplusplus
}
#endif
#endif
