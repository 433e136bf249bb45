/*
 * gpumon.h — stable C ABI of the MI355X-native GPU monitoring engine.
 *
 * One shared library (libgpumon.so) carries the device layer (amdsmi through a
 * dlopen'd function table, or a scripted fake), the host engine (field groups,
 * watches, health, policies, per-process accounting, introspection), the
 * Prometheus exporter and the pod-attribution client.  Everything above it —
 * the CLIs, the Go packages under bindings/go, the Python package — calls only
 * what is declared here.
 *
 * Reference parity: this ABI has one entry point for every libdcgm call the
 * reference's Go package makes through cgo (bindings/go/dcgm/admin.go:82-194,
 * device_info.go:47-129, fields.go:26-64, gpu_group.go:19-36,
 * device_status.go:129, health.go:27-54, policy.go:277-392,
 * process_info.go:87-101, hostengine_status.go:20-38, topology.go:113) and for
 * every NVML query of bindings/go/nvml/bindings.go:60-714, re-expressed with
 * AMD semantics (xGMI instead of NVLink, RAS blocks instead of SBE/DBE
 * locations, bad pages instead of retired pages, partitions instead of display
 * and persistence modes).
 *
 * Conventions
 *  - Every function returns gm_status_t; GM_ST_OK is 0, errors are negative.
 *  - A value a device cannot report is *blank* (gm_value_t.status != 0), never a
 *    sentinel number (the reference leaks 0x7ffffff0-style sentinels,
 *    bindings/go/dcgm/utils.go:15-18).
 *  - Structs that may grow carry a `version` = sizeof(struct) | (N << 24), the
 *    same handshake as bindings/go/dcgm/utils.go:127-135; use GM_VERSION().
 *  - Memory sizes are uint64 MiB or bytes: 288 GB of HBM3E overflows 32-bit.
 */
#ifndef GPUMON_H_
#define GPUMON_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GM_API __attribute__((visibility("default")))

#define GM_MAX_GPUS 64
#define GM_MAX_XGMI_LINKS 8
#define GM_MAX_XCD 8
#define GM_MAX_HBM_STACKS 8
#define GM_STR_LEN 96
#define GM_MAX_FIELDS_PER_GROUP 160
#define GM_MAX_INCIDENTS 32
#define GM_MAX_PROCS 256
#define GM_MAX_EVENT_TS 16

#define GM_VERSION(type, n) ((uint32_t)(sizeof(type) | ((uint32_t)(n) << 24)))

/* ---------------------------------------------------------------- status -- */
typedef enum {
  GM_ST_OK = 0,
  GM_ST_BADPARAM = -1,
  GM_ST_GENERIC_ERROR = -3,
  GM_ST_MEMORY = -4,
  GM_ST_NOT_CONFIGURED = -5,
  GM_ST_NOT_SUPPORTED = -6,
  GM_ST_INIT_ERROR = -7,
  GM_ST_LIBRARY_NOT_FOUND = -8,
  GM_ST_FUNCTION_NOT_FOUND = -9,
  GM_ST_NO_PERMISSION = -10,
  GM_ST_UNINITIALIZED = -11,
  GM_ST_NOT_WATCHED = -12,
  GM_ST_NO_DATA = -13,
  GM_ST_STALE_DATA = -14,
  GM_ST_CONNECTION_NOT_VALID = -15,
  GM_ST_GPU_NOT_SUPPORTED = -16,
  GM_ST_GROUP_INCOMPATIBLE = -17,
  GM_ST_MAX_LIMIT = -18,
  GM_ST_VER_MISMATCH = -19,
  GM_ST_TIMEOUT = -20,
  GM_ST_DUPLICATE_KEY = -21,
  GM_ST_IN_USE = -22,
  GM_ST_NOT_FOUND = -23,
  GM_ST_CHILD_SPAWN_FAILED = -24,
  GM_ST_INSUFFICIENT_SIZE = -25
} gm_status_t;

GM_API const char* gm_strerror(gm_status_t st);

/* ------------------------------------------------------------ field model -- */
typedef enum { GM_FT_INT64 = 1, GM_FT_DOUBLE = 2, GM_FT_STRING = 3 } gm_field_type_t;

/* Why a value is blank. */
typedef enum {
  GM_VS_OK = 0,
  GM_VS_BLANK = 1,          /* never sampled yet */
  GM_VS_NOT_SUPPORTED = 2,  /* device / driver does not expose it */
  GM_VS_NO_PERMISSION = 3,  /* needs privileges the engine lacks */
  GM_VS_ERROR = 4           /* the backend call failed this tick */
} gm_value_status_t;

/*
 * Field identifiers.  Names and meaning are AMD-native (gfx clock, UMC
 * activity, RAS correctable/uncorrectable, bad pages, xGMI).  The numbers of
 * the fields that back the Prometheus contract coincide with the ids operators
 * already pass to `dmon -e` (exporters/prometheus-dcgm/dcgm-exporter/
 * dcgm-exporter:85-95) so existing field lists keep working.
 */
enum {
  /* identity / static attributes */
  GM_FI_DRIVER_VERSION = 1,
  GM_FI_DEV_NAME = 50,
  GM_FI_DEV_BRAND = 51,
  GM_FI_DEV_SERIAL = 53,
  GM_FI_DEV_UUID = 54,
  GM_FI_DEV_PCI_BDF = 57,
  GM_FI_DEV_VBIOS_VERSION = 85,
  GM_FI_DEV_FIRMWARE_VERSION = 80, /* inforom-image analog: PSP/SMU bundle */
  GM_FI_DEV_GFX_TARGET = 58,       /* "gfx950" */
  GM_FI_DEV_DRM_RENDER_MINOR = 59,
  GM_FI_DEV_NUMA_NODE = 60,
  GM_FI_DEV_CPU_AFFINITY = 61,
  GM_FI_DEV_NUM_XCD = 62,
  GM_FI_DEV_NUM_CU = 63,
  GM_FI_DEV_COMPUTE_PARTITION = 64, /* SPX/DPX/QPX/CPX */
  GM_FI_DEV_MEMORY_PARTITION = 65,  /* NPS1/NPS2/... */
  GM_FI_DEV_HIP_ID = 66,
  GM_FI_DEV_OAM_ID = 67,
  GM_FI_DEV_PARTITION_ID = 68,      /* index of this compute partition inside its GPU (0 when not partitioned) */
  GM_FI_DEV_PCIE_MAX_GEN = 235,
  GM_FI_DEV_PCIE_MAX_WIDTH = 236,
  GM_FI_DEV_PCIE_GEN = 237,
  GM_FI_DEV_PCIE_WIDTH = 238,
  GM_FI_DEV_POWER_CAP = 160,         /* W */
  GM_FI_DEV_POWER_CAP_DEFAULT = 163, /* W */
  GM_FI_DEV_POWER_CAP_MAX = 162,     /* W */
  GM_FI_DEV_GFX_CLOCK_MAX = 113,     /* MHz */
  GM_FI_DEV_MEM_CLOCK_MAX = 114,     /* MHz */
  GM_FI_DEV_VIS_VRAM_TOTAL = 90,     /* MiB, host-visible aperture (BAR1 analog) */
  GM_FI_DEV_VIS_VRAM_USED = 92,      /* MiB */
  GM_FI_DEV_GTT_USED = 93,           /* MiB */

  /* clocks */
  GM_FI_GFX_CLOCK = 100, /* MHz, mean over active XCDs */
  GM_FI_MEM_CLOCK = 101, /* MHz, UCLK */
  GM_FI_SOC_CLOCK = 102,
  GM_FI_VCLK = 103,
  GM_FI_DCLK = 104,

  /* temperatures, °C */
  GM_FI_MEM_TEMP = 140, /* max over HBM stacks */
  GM_FI_GPU_TEMP = 150, /* hotspot / junction */
  GM_FI_EDGE_TEMP = 151,
  GM_FI_VR_GFX_TEMP = 152,
  GM_FI_VR_SOC_TEMP = 153,
  GM_FI_VR_MEM_TEMP = 154,
  GM_FI_HBM_STACK_TEMP_0 = 141, /* ..148, one per HBM stack */

  /* power / energy */
  GM_FI_POWER_USAGE = 155,  /* W, socket power */
  GM_FI_TOTAL_ENERGY = 156, /* mJ since driver load */

  /* PCIe */
  GM_FI_PCIE_TX_KB = 200, /* cumulative KB sent */
  GM_FI_PCIE_RX_KB = 201,
  GM_FI_PCIE_REPLAY_COUNTER = 202,
  GM_FI_PCIE_BW_INST = 208,      /* MB/s instantaneous */
  GM_FI_PCIE_L0_TO_RECOVERY = 209,
  GM_FI_PCIE_NAK_SENT = 210,
  GM_FI_PCIE_NAK_RCVD = 211,

  /* utilisation, % */
  GM_FI_GFX_UTIL = 203,
  GM_FI_MEM_UTIL = 204, /* UMC (HBM controller) activity */
  GM_FI_ENC_UTIL = 206, /* always not-supported on MI355X: no encoder */
  GM_FI_DEC_UTIL = 207, /* VCN / JPEG activity */

  /* events and throttling */
  GM_FI_LAST_EVENT_CODE = 230,       /* XID analog: last critical driver event */
  GM_FI_POWER_VIOLATION = 240,       /* us throttled by package power tracking */
  GM_FI_THERMAL_VIOLATION = 241,     /* us throttled by socket + HBM thermal */
  GM_FI_SYNC_BOOST_VIOLATION = 242,  /* no AMD meaning: never supported */
  GM_FI_BOARD_LIMIT_VIOLATION = 243, /* us: PROCHOT + VR thermal */
  GM_FI_LOW_UTIL_VIOLATION = 244,
  GM_FI_RELIABILITY_VIOLATION = 245, /* no AMD counterpart: blank on real devices (see GM_FI_GFX_BELOW_HOST_LIMIT) */
  GM_FI_THROTTLE_STATUS = 246,       /* bitset, see GM_THROTTLE_* */
  GM_FI_GFX_BELOW_HOST_LIMIT = 247,  /* us the shader clock spent under the limit the host set, whatever the cause:
                                        an idle GPU accrues it at wall-clock rate, so it is no "violation" */
  GM_FI_PERF_LEVEL = 190,            /* P-state analog: DPM level, 0 = highest */
  GM_FI_FAN_SPEED = 191,             /* % (OAM modules: not supported) */

  /* HBM capacity, MiB */
  GM_FI_VRAM_TOTAL = 250,
  GM_FI_VRAM_FREE = 251,
  GM_FI_VRAM_USED = 252,

  /* RAS (ECC) */
  GM_FI_ECC_CE_VOLATILE = 310, /* correctable since driver load */
  GM_FI_ECC_UE_VOLATILE = 311, /* uncorrectable since driver load */
  GM_FI_ECC_CE_AGGREGATE = 312,
  GM_FI_ECC_UE_AGGREGATE = 313,
  GM_FI_ECC_DEFERRED = 314,
  GM_FI_ECC_UMC_CE = 320, /* per RAS block */
  GM_FI_ECC_UMC_UE = 321,
  GM_FI_ECC_GFX_CE = 322,
  GM_FI_ECC_GFX_UE = 323,
  GM_FI_ECC_SDMA_CE = 324,
  GM_FI_ECC_SDMA_UE = 325,
  GM_FI_ECC_MMHUB_CE = 326,
  GM_FI_ECC_MMHUB_UE = 327,
  GM_FI_ECC_XGMI_CE = 328,
  GM_FI_ECC_XGMI_UE = 329,

  /* bad (retired) HBM pages */
  GM_FI_BAD_PAGES_CE = 390,      /* reserved because of correctable storms */
  GM_FI_BAD_PAGES_UE = 391,      /* reserved because of uncorrectable errors */
  GM_FI_BAD_PAGES_PENDING = 392,

  /* xGMI */
  GM_FI_XGMI_CRC_ERRORS = 409,      /* correctable link errors (XGMI_WAFL block) */
  GM_FI_XGMI_DATA_ERRORS = 419,     /* uncorrectable link errors */
  GM_FI_XGMI_REPLAY_ERRORS = 429,   /* not exposed by the driver: blank */
  GM_FI_XGMI_RECOVERY_ERRORS = 439, /* not exposed by the driver: blank */
  GM_FI_XGMI_BANDWIDTH_TOTAL = 449, /* KB, sum of rx+tx accumulators, all links */
  GM_FI_XGMI_LINK_RX_KB_0 = 450,    /* ..457 per-link cumulative KB read */
  GM_FI_XGMI_LINK_TX_KB_0 = 460,    /* ..467 per-link cumulative KB written */
  GM_FI_XGMI_LINK_STATUS_0 = 470,   /* ..477 1 = up, 0 = down */
  GM_FI_XGMI_LINK_WIDTH = 478,
  GM_FI_XGMI_LINK_SPEED = 479, /* Gbps per lane */
  GM_FI_XGMI_LINKS_UP = 480,
  GM_FI_XGMI_LINKS_DOWN = 481,
  GM_FI_XGMI_ERROR_STATUS = 482, /* 0 no errors, 1 an error, 2 several errors latched by the link layer */

  /* profiling-class metrics (`-p`), % */
  GM_FI_PROF_GR_ENGINE_ACTIVE = 1001,
  GM_FI_PROF_CU_ACTIVE = 1002,
  GM_FI_PROF_CU_OCCUPANCY = 1003,
  GM_FI_PROF_MFMA_ACTIVE = 1004,
  GM_FI_PROF_HBM_ACTIVE = 1005,
  GM_FI_PROF_MEM_PIPE_ACTIVE = 1020, /* vector-memory address units busy, % (hardware counters only) */
  GM_FI_XCD_BUSY_0 = 1100, /* ..1107 per-XCD gfx busy % */

  GM_FI_MAX_ID = 1200
};

/* Bits of GM_FI_THROTTLE_STATUS (a bitset, not an exact-match enum: the
 * reference's exact match loses composite masks, bindings/go/nvml/bindings.go:597-619). */
#define GM_THROTTLE_POWER (1u << 0)        /* package power tracking (PPT) */
#define GM_THROTTLE_THERMAL (1u << 1)      /* socket thermal */
#define GM_THROTTLE_HBM_THERMAL (1u << 2)
#define GM_THROTTLE_VR_THERMAL (1u << 3)
#define GM_THROTTLE_PROCHOT (1u << 4)
#define GM_THROTTLE_LOW_UTIL (1u << 5)
#define GM_THROTTLE_HOST_LIMIT (1u << 6)   /* gfx clock below host limit */

typedef struct {
  uint16_t field_id;
  uint8_t type;   /* gm_field_type_t */
  uint8_t status; /* gm_value_status_t; anything but GM_VS_OK means blank */
  uint32_t gpu;
  int64_t ts_us; /* wall-clock microseconds of the sample */
  union {
    int64_t i64;
    double dbl;
    char str[GM_STR_LEN];
  } v;
} gm_value_t;

typedef struct {
  uint16_t field_id;
  uint8_t type;
  uint8_t is_static;  /* sampled once at start-up */
  uint8_t prom_type;  /* 0 = not exported, 1 = gauge, 2 = counter */
  uint8_t prom_class; /* 0 = contract set, 1 = profiling set (-p), 2 = AMD extras */
  uint16_t reserved;
  const char* tag;       /* short AMD-native name, e.g. "gfx_clock" */
  const char* unit;      /* "MHz", "W", ... */
  const char* prom_name; /* full exposition name, e.g. "dcgm_sm_clock" */
  const char* prom_help;
} gm_field_meta_t;

GM_API gm_status_t gm_field_meta(uint16_t field_id, gm_field_meta_t* out);
GM_API gm_status_t gm_field_by_tag(const char* tag, uint16_t* field_id);
/* Fills ids[] with every registered field id; *n in = capacity, out = count. */
GM_API gm_status_t gm_field_list(uint16_t* ids, int* n);

/* --------------------------------------------------------------- lifecycle -- */
typedef enum {
  GM_MODE_EMBEDDED = 0,        /* engine threads live in this process */
  GM_MODE_STANDALONE = 1,      /* connect to a running gpumon-hostengine */
  GM_MODE_START_HOSTENGINE = 2 /* spawn gpumon-hostengine as a child, then connect */
} gm_mode_t;

typedef struct {
  uint32_t version;
  int32_t mode;           /* gm_mode_t */
  const char* address;    /* STANDALONE: "host[:port]" (default port 5555) or a unix socket path */
  int32_t address_is_unix_socket;
  const char* backend;    /* EMBEDDED: "amdsmi" (default), "fake"; env GPUMON_BACKEND overrides NULL */
  const char* backend_spec; /* fake: "gpus=8,seed=1,..." ; amdsmi: optional library path */
  const char* hostengine_path; /* START_HOSTENGINE: binary to spawn; NULL = search PATH */
} gm_init_params_t;
#define GM_INIT_PARAMS_VERSION GM_VERSION(gm_init_params_t, 1)

/* Reference-counted like bindings/go/dcgm/api.go:19-47: only the 0->1 call
 * initialises (later calls ignore params), only the 1->0 call tears down,
 * gm_shutdown with a zero count returns GM_ST_UNINITIALIZED. */
GM_API gm_status_t gm_init(const gm_init_params_t* params);
GM_API gm_status_t gm_shutdown(void);
GM_API int gm_init_refcount(void);

/* Serve the (embedded) engine of this process to remote clients.  address is
 * a unix socket path when is_unix != 0, else "host:port" / ":port". */
GM_API gm_status_t gm_server_start(const char* address, int is_unix);
GM_API gm_status_t gm_server_stop(void);

/* --------------------------------------------------------------- discovery -- */
GM_API gm_status_t gm_device_count(uint32_t* count);
/* GPUs the engine can fully sample (CDNA-class devices). */
GM_API gm_status_t gm_supported_devices(uint32_t* ids, int* n);

typedef struct {
  uint32_t version;
  uint32_t gpu;
  int32_t supported;
  char uuid[GM_STR_LEN];
  char name[GM_STR_LEN];
  char brand[GM_STR_LEN];
  char serial[GM_STR_LEN];
  char vbios[GM_STR_LEN];
  char firmware[GM_STR_LEN];
  char driver_version[GM_STR_LEN];
  char pci_bdf[GM_STR_LEN];
  char gfx_target[32];
  char device_path[64];       /* /dev/dri/renderD<N> */
  char cpu_affinity[GM_STR_LEN]; /* sysfs local_cpulist, e.g. "0-47,96-143" */
  char compute_partition[16];
  char memory_partition[16];
  int32_t numa_node;          /* -1 = unknown */
  int32_t hip_id;
  int32_t drm_render_minor;
  uint32_t num_xcd;
  uint32_t num_cu;
  uint32_t pcie_max_gen;
  uint32_t pcie_max_width;
  int64_t pcie_bandwidth_mbs; /* derived from gen x width; -1 = blank */
  double power_cap_w;         /* blank = NaN */
  double power_cap_default_w;
  int64_t vram_total_mib;
  int64_t vis_vram_total_mib;
  int64_t gfx_clock_max_mhz;
  int64_t mem_clock_max_mhz;
} gm_device_attributes_t;
#define GM_DEVICE_ATTRIBUTES_VERSION GM_VERSION(gm_device_attributes_t, 1)

GM_API gm_status_t gm_device_attributes(uint32_t gpu, gm_device_attributes_t* out);

/* Link class between two GPUs, ordered from farthest to closest like the
 * reference's P2PLinkType (bindings/go/nvml/nvml.go:131-147), with xGMI
 * replacing the NVLink counts. */
typedef enum {
  GM_LINK_UNKNOWN = 0,
  GM_LINK_CROSS_CPU = 1,     /* SYS  */
  GM_LINK_SAME_CPU = 2,      /* NODE */
  GM_LINK_HOST_BRIDGE = 3,   /* PHB  */
  GM_LINK_MULTI_SWITCH = 4,  /* PXB  */
  GM_LINK_SINGLE_SWITCH = 5, /* PIX  */
  GM_LINK_SAME_BOARD = 6,    /* PSB  */
  GM_LINK_XGMI = 7           /* XGMI, `hops` says how many */
} gm_link_type_t;

typedef struct {
  uint32_t gpu; /* peer */
  int32_t link; /* gm_link_type_t */
  uint32_t hops;
  uint32_t num_links;  /* parallel xGMI links to this peer */
  uint64_t weight;
  int64_t min_bw_mbs; /* -1 = blank */
  int64_t max_bw_mbs;
  int32_t p2p_accessible;
  char pci_bdf[GM_STR_LEN]; /* the PEER's bus id */
} gm_peer_link_t;

typedef struct {
  uint32_t version;
  uint32_t gpu;
  char cpu_affinity[GM_STR_LEN];
  int32_t numa_node;
  uint32_t num_peers;
  gm_peer_link_t peers[GM_MAX_GPUS];
} gm_topology_t;
#define GM_TOPOLOGY_VERSION GM_VERSION(gm_topology_t, 1)

GM_API gm_status_t gm_device_topology(uint32_t gpu, gm_topology_t* out);

/* One physical xGMI link of a GPU: its state, who sits at the other end, its rate and its traffic counters —
 * what the reference gets by walking nvmlDeviceGetNvLinkState / RemotePciInfo over the link indices
 * (bindings/go/nvml/bindings.go:268-318).  `link` is the index the per-link telemetry fields use
 * (GM_FI_XGMI_LINK_*_0 + link). */
typedef struct {
  uint32_t link;
  int32_t state;             /* 1 up, 0 down, -1 not wired / not reported */
  int32_t peer_gpu;          /* engine index of the GPU at the other end; -1 when it is not one of ours or unknown */
  char peer_bdf[GM_STR_LEN]; /* PCI address of the other end; empty when the platform does not say */
  int64_t bit_rate_gbps;     /* current rate; -1 = blank */
  int64_t max_bandwidth_gbps;
  int64_t rx_kb, tx_kb;      /* cumulative; -1 = blank */
} gm_xgmi_link_t;
/* *n in = capacity of out (GM_MAX_XGMI_LINKS is enough), out = links reported. */
GM_API gm_status_t gm_xgmi_links(uint32_t gpu, gm_xgmi_link_t* out, int* n);
GM_API const char* gm_link_type_name(int32_t link, uint32_t hops);  /* "XGMI1", "PIX", ... */
GM_API const char* gm_link_type_description(int32_t link);

/* ------------------------------------------------- groups / field groups -- */
typedef uint32_t gm_group_t;
typedef uint32_t gm_field_group_t;
#define GM_GROUP_ALL_GPUS ((gm_group_t)0x7fffffff)

GM_API gm_status_t gm_group_create(const char* name, gm_group_t* out); /* empty group */
GM_API gm_status_t gm_group_add_device(gm_group_t g, uint32_t gpu);
GM_API gm_status_t gm_group_remove_device(gm_group_t g, uint32_t gpu);
GM_API gm_status_t gm_group_devices(gm_group_t g, uint32_t* ids, int* n);
GM_API gm_status_t gm_group_destroy(gm_group_t g);

GM_API gm_status_t gm_field_group_create(const uint16_t* field_ids, int n, const char* name,
                                         gm_field_group_t* out);
GM_API gm_status_t gm_field_group_destroy(gm_field_group_t fg);

/* Defaults of the reference (bindings/go/dcgm/fields.go:12-16). */
#define GM_DEFAULT_UPDATE_FREQ_US 1000000LL
#define GM_DEFAULT_MAX_KEEP_AGE_S 300.0
#define GM_DEFAULT_MAX_KEEP_SAMPLES 0

GM_API gm_status_t gm_watch_fields(gm_group_t g, gm_field_group_t fg, int64_t update_freq_us,
                                   double max_keep_age_s, int32_t max_keep_samples);
GM_API gm_status_t gm_unwatch_fields(gm_group_t g, gm_field_group_t fg);
/* Forces one sample cycle of everything watched; wait != 0 blocks until done. */
GM_API gm_status_t gm_update_all_fields(int wait);

GM_API gm_status_t gm_latest_values(uint32_t gpu, const uint16_t* field_ids, int n,
                                    gm_value_t* out);
/* History of one field since ts_us (exclusive); *n in = capacity, out = count. */
GM_API gm_status_t gm_values_since(uint32_t gpu, uint16_t field_id, int64_t since_ts_us,
                                   gm_value_t* out, int* n);

/* Overwrite the cached value of (gpu, field) as if the device had reported it:
 * drives health and policy without breaking hardware.  The injected value is
 * held for hold_ms (0 = until the next real sample replaces it). */
GM_API gm_status_t gm_inject_value(uint32_t gpu, const gm_value_t* value, int32_t hold_ms);
/* Fake backend only: "gpu=2,field=150,value=97", "gpu=1,event=reset", "link_down=0:3", ... */
GM_API gm_status_t gm_fake_ctl(const char* command);

/* ---------------------------------------------------------- device status -- */
/* Optional values use a parallel validity mask instead of pointers. */
typedef struct {
  uint32_t version;
  uint32_t gpu;
  uint64_t valid; /* bit i set = member i (order below) holds a value */
  double power_w;            /* 0 */
  int64_t temperature_c;     /* 1 */
  int64_t gfx_util;          /* 2 */
  int64_t mem_util;          /* 3 */
  int64_t enc_util;          /* 4 */
  int64_t dec_util;          /* 5 */
  int64_t gfx_clock_mhz;     /* 6 */
  int64_t mem_clock_mhz;     /* 7 */
  int64_t vis_vram_used_mib; /* 8  (BAR1 used analog) */
  int64_t pcie_rx_kb;        /* 9 */
  int64_t pcie_tx_kb;        /* 10 */
  int64_t pcie_replays;      /* 11 */
  int64_t vram_used_mib;     /* 12 */
  int64_t vram_free_mib;     /* 13 */
  int64_t ecc_ce;            /* 14 */
  int64_t ecc_ue;            /* 15 */
  int64_t perf_level;        /* 16 */
  int64_t fan_speed;         /* 17 */
  int64_t throttle_status;   /* 18 bitset */
  int64_t mem_temp_c;        /* 19 */
  int64_t energy_mj;         /* 20 */
} gm_device_status_t;
#define GM_DEVICE_STATUS_VERSION GM_VERSION(gm_device_status_t, 1)
enum {
  GM_DS_POWER = 0, GM_DS_TEMP, GM_DS_GFX_UTIL, GM_DS_MEM_UTIL, GM_DS_ENC_UTIL, GM_DS_DEC_UTIL,
  GM_DS_GFX_CLOCK, GM_DS_MEM_CLOCK, GM_DS_VIS_VRAM_USED, GM_DS_PCIE_RX, GM_DS_PCIE_TX,
  GM_DS_PCIE_REPLAYS, GM_DS_VRAM_USED, GM_DS_VRAM_FREE, GM_DS_ECC_CE, GM_DS_ECC_UE,
  GM_DS_PERF_LEVEL, GM_DS_FAN, GM_DS_THROTTLE, GM_DS_MEM_TEMP, GM_DS_ENERGY
};

/* Served from the sampler cache through one persistent internal watch; no
 * per-call group churn (the reference creates and destroys a group and a
 * field group on every call, bindings/go/dcgm/device_status.go:115-180). */
GM_API gm_status_t gm_device_status(uint32_t gpu, gm_device_status_t* out);

/* Processes currently holding the GPU (direct device query, not accounting).  The list comes from the device
 * library's process table where the host has one (source "kfd"); else from the kernel's per-file DRM statistics
 * under /proc/<pid>/fdinfo ("fdinfo": memory and engine time of every process the caller may look into); else
 * from the open render-node descriptors under /proc/<pid>/fd ("open-files": who holds the card, without figures:
 * vram_bytes and gtt_bytes are then GM_PROC_MEM_UNKNOWN). */
#define GM_PROC_MEM_UNKNOWN (~0ull)
typedef struct {
  uint32_t pid;
  uint32_t type; /* 0 = compute, 1 = graphics, 2 = both */
  uint64_t vram_bytes; /* GM_PROC_MEM_UNKNOWN = not known */
  uint64_t gtt_bytes;
  uint64_t gfx_engine_ns;  /* cumulative gfx engine time */
  uint32_t cu_occupancy;   /* CUs currently occupied; 0xffffffff = blank */
  char name[GM_STR_LEN];
} gm_process_t;
GM_API gm_status_t gm_device_processes(uint32_t gpu, gm_process_t* out, int* n);
/* Which source answers gm_device_processes for this GPU on this host: "kfd", "fdinfo", "open-files", "fake" or
 * "none" (then gm_device_processes reports NOT_SUPPORTED or NO_PERMISSION). */
GM_API gm_status_t gm_process_source(uint32_t gpu, char* buf, size_t cap);
/* The /proc walk on its own (no engine, no device library): users of the card at `pci_bdf` / render node
 * `render_minor` under `proc_root` ("/proc").  *n: capacity on entry, count on return; `source` as above. */
GM_API gm_status_t gm_proc_scan(const char* proc_root, const char* pci_bdf, int render_minor, gm_process_t* out, int* n,
                                char* source, size_t cap);
/* The amdgpu driver's bulk metrics table (sysfs gpu_metrics) decoded without the device library: `json` gets the
 * members the sampler uses, all-ones for what the table does not report.  NOT_SUPPORTED (and the reason in `json`)
 * for a table layout other than format 1 content 8; the engine then keeps to the device library. */
GM_API gm_status_t gm_metrics_table_decode(const void* raw, size_t n, char* json, size_t cap);
/* How the amdsmi backend reads that table for this GPU at the moment: "direct" (own read of the sysfs file, checked
 * against the library) or "library"; with the reason where direct reads were refused. */
GM_API gm_status_t gm_metrics_table_source(uint32_t gpu, char* buf, size_t cap);

typedef struct {
  uint32_t version;
  uint32_t gpu;
  int32_t accounting_enabled; /* engine PID watch active on this GPU */
  int32_t accounting_buffer_size;
  char compute_partition[16];
  char memory_partition[16];
} gm_device_mode_t;
#define GM_DEVICE_MODE_VERSION GM_VERSION(gm_device_mode_t, 1)
GM_API gm_status_t gm_device_mode(uint32_t gpu, gm_device_mode_t* out);

/* ------------------------------------------------------------------ health -- */
#define GM_HEALTH_WATCH_PCIE (1u << 0)
#define GM_HEALTH_WATCH_XGMI (1u << 1)
#define GM_HEALTH_WATCH_MEM (1u << 2)      /* HBM ECC + bad pages */
#define GM_HEALTH_WATCH_COMPUTE (1u << 3)  /* XCD / GFX RAS */
#define GM_HEALTH_WATCH_THERMAL (1u << 4)
#define GM_HEALTH_WATCH_POWER (1u << 5)
#define GM_HEALTH_WATCH_DRIVER (1u << 6)   /* resets, VM faults, RAS events */
#define GM_HEALTH_WATCH_FIRMWARE (1u << 7)
#define GM_HEALTH_WATCH_ALL 0xffu

typedef enum { GM_HEALTH_PASS = 0, GM_HEALTH_WARN = 10, GM_HEALTH_FAIL = 20 } gm_health_result_t;

typedef struct {
  uint32_t system; /* one GM_HEALTH_WATCH_* bit */
  int32_t health;  /* gm_health_result_t */
  char message[192];
} gm_incident_t;

typedef struct {
  uint32_t version;
  uint32_t gpu;
  int32_t overall; /* gm_health_result_t */
  uint32_t incident_count;
  gm_incident_t incidents[GM_MAX_INCIDENTS];
} gm_health_response_t;
#define GM_HEALTH_RESPONSE_VERSION GM_VERSION(gm_health_response_t, 1)

GM_API gm_status_t gm_health_set(gm_group_t g, uint32_t systems);
GM_API gm_status_t gm_health_get(gm_group_t g, uint32_t* systems);
/* One response per GPU of the group; *n in = capacity, out = count. */
GM_API gm_status_t gm_health_check(gm_group_t g, gm_health_response_t* out, int* n);

/* Levels the health checks compare against.  Defaults suit MI355X-class parts; every daemon takes them as flags
 * (--health-hotspot-warn ...), from its config file and from $GPUMON_ARGS. */
typedef struct {
  uint32_t version;
  int32_t hotspot_warn_c, hotspot_fail_c; /* 100 / 110 */
  int32_t mem_warn_c, mem_fail_c;         /* 95 / 105 */
  int32_t bad_pages_fail;                 /* 256: the RAS bad-page table is nearly full */
  int32_t ce_burst_warn;                  /* 100 correctable errors between two checks */
  double power_over_cap_warn;             /* 1.10: socket power above cap x this */
} gm_health_thresholds_t;
#define GM_HEALTH_THRESHOLDS_VERSION GM_VERSION(gm_health_thresholds_t, 1)
GM_API gm_status_t gm_health_thresholds_set(const gm_health_thresholds_t* t);
GM_API gm_status_t gm_health_thresholds_get(gm_health_thresholds_t* t);

/* Result of an active diagnostic (gpumon-diag: memtest, HBM stream, MFMA / GEMM stress, link tests) handed to the
 * engine.  A failed or degraded result stays with the GPU until a later result of the same test replaces it: health
 * checks report it under `system`, and a FAIL also raises a GM_EVENT_DIAG_FAIL event (GPU-event policy,
 * GM_FI_LAST_EVENT_CODE). */
typedef struct {
  uint32_t version;
  uint32_t gpu;
  char test[32];    /* "memtest", "hbm_stream", "gemm_bf16", "xgmi_p2p", ... */
  uint32_t system;  /* one GM_HEALTH_WATCH_* bit the result belongs to */
  int32_t health;   /* gm_health_result_t */
  double value;     /* the number the verdict rests on (errors, TB/s, TFLOP/s, GB/s); NaN when there is none */
  int64_t ts_us;    /* when the test finished; 0 = now */
  char message[192];
} gm_diag_result_t;
#define GM_DIAG_RESULT_VERSION GM_VERSION(gm_diag_result_t, 1)
/* Results are kept per (GPU, test name).  A name is made of [A-Za-z0-9_.-]; once a GPU has this many, a result under
 * a new name is refused with GM_ST_INSUFFICIENT_SIZE (results under known names keep replacing their predecessors). */
#define GM_MAX_DIAG_TESTS 32
GM_API gm_status_t gm_diag_result_submit(const gm_diag_result_t* r);
/* Latest result per (GPU, test); *n in = capacity, out = count. */
GM_API gm_status_t gm_diag_results(uint32_t gpu, gm_diag_result_t* out, int* n);
GM_API const char* gm_health_system_name(uint32_t system);
GM_API const char* gm_health_result_name(int32_t health);

/* ---------------------------------------------------------------- policies -- */
#define GM_POLICY_COND_ECC_UE (1u << 0)         /* uncorrectable RAS error */
#define GM_POLICY_COND_PCIE (1u << 1)           /* PCIe replay */
#define GM_POLICY_COND_MAX_BAD_PAGES (1u << 2)  /* bad-page count >= threshold */
#define GM_POLICY_COND_THERMAL (1u << 3)        /* hotspot >= threshold (°C) */
#define GM_POLICY_COND_POWER (1u << 4)          /* socket power >= threshold (W) */
#define GM_POLICY_COND_XGMI (1u << 5)           /* xGMI link error / link down */
#define GM_POLICY_COND_GPU_EVENT (1u << 6)      /* reset / VM fault / RAS: XID analog */
#define GM_POLICY_COND_ALL 0x7fu

/* Thresholds.  Defaults are sized for MI355X (1400 W board power class,
 * 105 °C hotspot throttle region); the reference hard-codes 10 pages /
 * 100 °C / 250 W (bindings/go/dcgm/policy.go:118-120). */
typedef struct {
  uint32_t version;
  uint32_t conditions;  /* mask of GM_POLICY_COND_* the thresholds apply to */
  uint32_t max_bad_pages;   /* default 10 */
  uint32_t thermal_c;       /* default 100 */
  uint32_t power_w;         /* default 1400 */
  uint32_t hysteresis_pct;  /* re-arm when the value falls this % below the threshold; default 3 */
} gm_policy_t;
#define GM_POLICY_VERSION GM_VERSION(gm_policy_t, 1)

typedef struct {
  uint32_t condition; /* exactly one GM_POLICY_COND_* bit */
  uint32_t gpu;
  int64_t ts_us;
  union {
    struct { uint32_t block; uint32_t count; char location[32]; } ecc;   /* ECC_UE */
    struct { uint32_t replay_counter; } pcie;                           /* PCIE */
    struct { uint32_t ce_pages; uint32_t ue_pages; } bad_pages;         /* MAX_BAD_PAGES */
    struct { uint32_t temperature_c; } thermal;                         /* THERMAL */
    struct { uint32_t power_w; } power;                                 /* POWER */
    struct { uint16_t field_id; uint16_t link; uint32_t counter; } xgmi;/* XGMI */
    struct { uint32_t code; char message[64]; } event;                  /* GPU_EVENT */
  } d;
} gm_violation_t;

/* Codes of GM_POLICY_COND_GPU_EVENT / GM_FI_LAST_EVENT_CODE. */
enum {
  GM_EVENT_NONE = 0,
  GM_EVENT_VMFAULT = 1,
  GM_EVENT_THERMAL_THROTTLE = 2,
  GM_EVENT_GPU_PRE_RESET = 3,
  GM_EVENT_GPU_POST_RESET = 4,
  GM_EVENT_RAS_UE = 5,
  GM_EVENT_RAS_POISON = 6,
  GM_EVENT_RING_HANG = 7,
  GM_EVENT_DIAG_FAIL = 8, /* an active diagnostic failed on this GPU (gm_diag_result_submit) */
  GM_EVENT_PROCESS_START = 20, /* not critical: never a violation */
  GM_EVENT_PROCESS_END = 21
};

/* Called on the engine's notifier thread, never on the sampler thread. */
typedef int (*gm_violation_cb)(const gm_violation_t* v, void* user);
typedef uint32_t gm_policy_reg_t;

GM_API gm_status_t gm_policy_set(gm_group_t g, const gm_policy_t* p);
GM_API gm_status_t gm_policy_get(gm_group_t g, gm_policy_t* p);
GM_API gm_status_t gm_policy_register(gm_group_t g, uint32_t conditions, gm_violation_cb cb,
                                      void* user, gm_policy_reg_t* out);
GM_API gm_status_t gm_policy_unregister(gm_policy_reg_t reg);
/* Pull-style alternative for callers that cannot take callbacks (ctypes, Go
 * without cgo callbacks): blocks up to timeout_ms for the next violation of
 * this registration.  cb may be NULL at registration for pull-only use. */
GM_API gm_status_t gm_policy_wait(gm_policy_reg_t reg, int32_t timeout_ms, gm_violation_t* out);
GM_API const char* gm_policy_condition_name(uint32_t condition);

/* ---------------------------------------------------- per-process accounting -- */
typedef struct {
  uint32_t gpu;
  uint32_t pid;
  uint64_t valid; /* bit per member below, in order, starting at energy_mj */
  int64_t start_ts_us;
  int64_t end_ts_us; /* 0 = still running */
  int64_t energy_mj;         /* bit 0: energy attributed to the process */
  double gfx_util_avg;       /* bit 1: device gfx util averaged over the process lifetime */
  double mem_util_avg;       /* bit 2 */
  double proc_gfx_util_avg;  /* bit 3: the process' own share of the gfx engine */
  double proc_mem_util_avg;  /* bit 4 */
  int64_t pcie_rx_avg_kbs;   /* bit 5 */
  int64_t pcie_tx_avg_kbs;   /* bit 6 */
  int64_t pcie_replays;      /* bit 7 */
  int64_t max_vram_bytes;    /* bit 8 */
  int64_t ecc_ce;            /* bit 9 */
  int64_t ecc_ue;            /* bit 10 */
  int64_t gfx_clock_avg;     /* bit 11 */
  int64_t mem_clock_avg;     /* bit 12 */
  int64_t power_violation_us;       /* bit 13 */
  int64_t thermal_violation_us;     /* bit 14 */
  int64_t reliability_violation_us; /* bit 15 */
  int64_t board_limit_violation_us; /* bit 16 */
  int64_t low_util_violation_us;    /* bit 17 */
  int64_t sync_boost_violation_us;  /* bit 18: never valid on AMD */
  uint32_t num_critical_events;
  int64_t critical_event_ts_us[GM_MAX_EVENT_TS];
  char name[GM_STR_LEN];
} gm_pid_gpu_stats_t;

typedef struct {
  uint32_t version;
  uint32_t pid;
  uint32_t num_gpus;
  gm_pid_gpu_stats_t gpus[GM_MAX_GPUS];
} gm_pid_info_t;
#define GM_PID_INFO_VERSION GM_VERSION(gm_pid_info_t, 1)

GM_API gm_status_t gm_watch_pid_fields(gm_group_t g, int64_t update_freq_us, double max_keep_age_s,
                                       int32_t max_keep_samples);
GM_API gm_status_t gm_pid_info(gm_group_t g, uint32_t pid, gm_pid_info_t* out);

/* ----------------------------------------------------------- introspection -- */
typedef struct {
  uint32_t version;
  int64_t memory_kb;   /* resident set of the engine process */
  double cpu_percent;  /* engine process CPU since introspection was enabled, % of one core */
  int64_t uptime_us;
  uint64_t sample_cycles;
  uint64_t backend_errors;
  double last_cycle_ms;  /* wall time of the last sample cycle */
  double avg_cycle_ms;
} gm_introspect_t;
#define GM_INTROSPECT_VERSION GM_VERSION(gm_introspect_t, 1)
GM_API gm_status_t gm_introspect_toggle(int enabled);
GM_API gm_status_t gm_introspect(gm_introspect_t* out);

/* ------------------------------------------------------------------ events -- */
typedef struct {
  uint32_t gpu;
  uint32_t code; /* GM_EVENT_* */
  int64_t ts_us;
  char uuid[GM_STR_LEN];
  char message[96];
} gm_event_t;
/* Mirrors the NVML event-set API (bindings/go/nvml/bindings.go:68-146). */
typedef uint32_t gm_event_set_t;
GM_API gm_status_t gm_event_set_create(gm_event_set_t* out);
GM_API gm_status_t gm_event_register(gm_event_set_t es, uint32_t gpu /* or GM_MAX_GPUS = all */,
                                     uint64_t event_mask /* 1<<code, 0 = all critical */);
GM_API gm_status_t gm_event_wait(gm_event_set_t es, int32_t timeout_ms, gm_event_t* out);
GM_API gm_status_t gm_event_set_free(gm_event_set_t es);

/* ---------------------------------------------------------------- exporter -- */
typedef struct {
  uint32_t version;
  int32_t profiling;        /* -p: add the five profiling metrics */
  int32_t amd_extras;       /* add per-link xGMI, per-stack HBM, per-XCD families */
  int32_t self_metrics;     /* add gpumon_* exporter/engine self metrics */
  int64_t interval_ms;      /* -d, clamped to >= 100 */
  const char* gpus;         /* "0,2,5" or NULL/"-1" = all */
  const char* output_path;  /* -o; NULL = no file sink */
  const char* listen;       /* "host:port" / ":9400"; NULL = no HTTP */
  const char* kubelet_socket; /* non-NULL enables pod attribution */
  const char* resource_names; /* comma list, default "amd.com/gpu" */
  const char* pod_map_path;   /* file-based pod map fallback (device-plugin checkpoint) */
  int32_t rest_api;         /* serve the REST routes on the same listener */
  int32_t legacy_layout;    /* 1 = GPU-major layout of the reference's awk script */
  const char* diag_results_path; /* gpumon-diag --publish report (file, or directory of *.json): exported as gpumon_diag_* */
  const char* state_path;   /* NULL = none; file that carries what the exporter itself accumulates (the
                               gpu_last_not_idle_time latch, per GPU UUID) across restarts */
  int32_t health;           /* run the engine's health check on every sweep: gpumon_health_* families */
  int32_t slow_interval_ms; /* 0 = 1000.  Sources that cost a device-library call each and change slowly (error counters,
                               bad pages, performance level / power cap, memory usage) keep this period when
                               interval_ms is shorter */
  int32_t pod_refresh_ms;   /* 0 = 10000.  Period of the pod-map query (kubelet or file), on a thread of its own over
                               one kept connection: a kubelet that does not answer never delays a sweep */
} gm_exporter_config_t;
#define GM_EXPORTER_CONFIG_VERSION GM_VERSION(gm_exporter_config_t, 5)

typedef uint32_t gm_exporter_t;
GM_API gm_status_t gm_exporter_create(const gm_exporter_config_t* cfg, gm_exporter_t* out);
/* Render the current snapshot; returns bytes needed in *len (copy is truncated to capacity). */
GM_API gm_status_t gm_exporter_render(gm_exporter_t e, char* buf, size_t* len);
/* One sweep: force-sample, render, publish atomically to output_path. */
GM_API gm_status_t gm_exporter_sweep(gm_exporter_t e);
GM_API gm_status_t gm_exporter_start(gm_exporter_t e); /* background: sweeps + HTTP */
GM_API gm_status_t gm_exporter_stop(gm_exporter_t e);
GM_API gm_status_t gm_exporter_http_port(gm_exporter_t e, int* port);
GM_API gm_status_t gm_exporter_destroy(gm_exporter_t e);

/* Pod attribution building blocks (testable without an exporter). */
/* Rewrites exposition text: samples whose uuid (2nd label) or device id
 * "<prefix><gpu index>" is in the map get pod_name / pod_namespace /
 * container_name labels appended.  map lines: "<device-id>\t<pod>\t<namespace>\t<container>\n". */
GM_API gm_status_t gm_pod_relabel(const char* text, const char* map_tsv, const char* index_prefix,
                                  char* out, size_t* len);
/* Queries the kubelet PodResources API over its unix socket and returns the map as TSV. */
GM_API gm_status_t gm_kubelet_pod_map(const char* socket_path, const char* resource_names,
                                      int32_t timeout_ms, char* out, size_t* len);

/* File-watch pod exporter: watches source_path, rewrites it with pod labels into
 * dest_path on every (re)creation, serves dest_path at /gpu/metrics.  Needs no
 * engine (it only reads a metrics file and talks to the kubelet). */
typedef struct {
  uint32_t version;
  const char* source_path;    /* default /run/prometheus/dcgm.prom */
  const char* dest_path;      /* default /run/dcgm/dcgm-pod.prom */
  const char* kubelet_socket; /* default /var/lib/kubelet/pod-resources/kubelet.sock */
  const char* resource_names; /* default amd.com/gpu */
  const char* index_prefix;   /* device ids of the form <prefix><gpu index>; default amdgpu */
  const char* listen;         /* default :9400; "" = no HTTP */
  int32_t timeout_ms;         /* kubelet + HTTP timeouts; default 10000 */
  int32_t liveness_ms;        /* give up after this long without file events; default 600000 */
  int32_t pod_refresh_ms;     /* 0 = ask the kubelet on every sweep */
} gm_pod_exporter_config_t;
#define GM_POD_EXPORTER_CONFIG_VERSION GM_VERSION(gm_pod_exporter_config_t, 1)
typedef uint32_t gm_pod_exporter_t;
GM_API gm_status_t gm_pod_exporter_start(const gm_pod_exporter_config_t* cfg, gm_pod_exporter_t* out);
/* alive: 0 once the liveness window expired; port: bound HTTP port; rewrites: files published. */
GM_API gm_status_t gm_pod_exporter_status(gm_pod_exporter_t pe, int* alive, int* port, uint64_t* rewrites);
GM_API gm_status_t gm_pod_exporter_stop(gm_pod_exporter_t pe);

/* ------------------------------------------------------------------- views -- */
/* Ready-made text / JSON renderings (the layouts of the sample tools and the
 * REST routes) so every binding prints the same thing. */
typedef enum {
  GM_VIEW_DEVICE_INFO = 0, /* id = gpu */
  GM_VIEW_DEVICE_STATUS = 1,
  GM_VIEW_HEALTH = 2,
  GM_VIEW_PROCESS = 3,     /* id = pid */
  GM_VIEW_HOSTENGINE = 4,  /* id ignored */
  GM_VIEW_TOPOLOGY = 5,    /* id ignored; text only */
  GM_VIEW_LINKS = 6,       /* id = gpu: per-link xGMI table */
  GM_VIEW_DIAG = 7         /* id = gpu: latest verdict of every active diagnostic */
} gm_view_t;
GM_API gm_status_t gm_render_view(int view, uint32_t id, int json, char* out, size_t* len);
GM_API gm_status_t gm_format_violation(const gm_violation_t* v, char* out, size_t* len);
GM_API const char* gm_event_code_name(uint32_t code);

/* ------------------------------------------------------ direct device layer -- */
/* The "smi" path: every call goes straight to the device library, with no host
 * engine, no sampler thread and no cache — the role the NVML package plays in the
 * reference next to its DCGM package (bindings/go/nvml/nvml.go:274-604).  It has
 * its own refcounted lifetime, independent of gm_init. */
GM_API gm_status_t gm_smi_init(const char* backend /* NULL = "amdsmi" */, const char* backend_spec);
GM_API gm_status_t gm_smi_shutdown(void);
GM_API gm_status_t gm_smi_device_count(uint32_t* count);
GM_API gm_status_t gm_smi_driver_version(char* buf, size_t cap);
/* Version of the user-space stack: the ROCm release the device library belongs to ("7.2.0"), else the
 * library's own version; buf may be NULL.  The counterpart of nvml.GetCudaDriverVersion. */
GM_API gm_status_t gm_smi_runtime_version(uint32_t* major, uint32_t* minor, char* buf, size_t cap);
GM_API gm_status_t gm_smi_device_attributes(uint32_t gpu, gm_device_attributes_t* out);
/* One synchronous read of everything gm_device_status_t holds. */
GM_API gm_status_t gm_smi_device_status(uint32_t gpu, gm_device_status_t* out);
/* One synchronous read of arbitrary fields (one device-library round per source). */
GM_API gm_status_t gm_smi_read_fields(uint32_t gpu, const uint16_t* field_ids, int n, gm_value_t* out);
GM_API gm_status_t gm_smi_device_processes(uint32_t gpu, gm_process_t* out, int* n);
GM_API gm_status_t gm_smi_peer_link(uint32_t gpu, uint32_t peer, gm_peer_link_t* out);
GM_API gm_status_t gm_smi_xgmi_links(uint32_t gpu, gm_xgmi_link_t* out, int* n);
GM_API gm_status_t gm_smi_device_mode(uint32_t gpu, gm_device_mode_t* out);
/* Blocks up to timeout_ms for the next driver event (reset, RAS, thermal...);
 * GM_ST_TIMEOUT when none arrived, GM_ST_NOT_SUPPORTED when the platform has no event channel. */
GM_API gm_status_t gm_smi_event_wait(int32_t timeout_ms, gm_event_t* out);
/* Scripts the fake backend owned by the direct layer (tests); NOT_SUPPORTED on real hardware. */
GM_API gm_status_t gm_smi_fake_ctl(const char* command);

/* ---------------------------------------------------------------- logging -- */
GM_API void gm_set_log_level(int level); /* 0 error, 1 warn, 2 info, 3 debug */

#ifdef __cplusplus
}
#endif
#endif /* GPUMON_H_ */
