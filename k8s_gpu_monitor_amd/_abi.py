"""ctypes mirror of include/gpumon.h.

Every versioned struct carries ``version = sizeof | (1 << 24)`` exactly like the
C macro GM_VERSION, so a drift between this file and the header is caught by
the library (GM_ST_VER_MISMATCH) instead of corrupting memory.
"""
import ctypes as C

GM_MAX_GPUS = 64
GM_MAX_XGMI_LINKS = 8
GM_STR_LEN = 96
GM_MAX_INCIDENTS = 32
GM_MAX_PROCS = 256
GM_MAX_EVENT_TS = 16
GM_GROUP_ALL_GPUS = 0x7FFFFFFF

# status codes
GM_ST_OK = 0
GM_ST_BADPARAM = -1
GM_ST_GENERIC_ERROR = -3
GM_ST_NOT_CONFIGURED = -5
GM_ST_NOT_SUPPORTED = -6
GM_ST_INIT_ERROR = -7
GM_ST_LIBRARY_NOT_FOUND = -8
GM_ST_UNINITIALIZED = -11
GM_ST_NOT_WATCHED = -12
GM_ST_NO_DATA = -13
GM_ST_CONNECTION_NOT_VALID = -15
GM_ST_GPU_NOT_SUPPORTED = -16
GM_ST_VER_MISMATCH = -19
GM_ST_TIMEOUT = -20
GM_ST_DUPLICATE_KEY = -21
GM_ST_NOT_FOUND = -23
GM_ST_CHILD_SPAWN_FAILED = -24
GM_ST_INSUFFICIENT_SIZE = -25
GM_ST_MEMORY = -4
GM_ST_FUNCTION_NOT_FOUND = -9
GM_ST_NO_PERMISSION = -10
GM_ST_STALE_DATA = -14
GM_ST_GROUP_INCOMPATIBLE = -17
GM_ST_MAX_LIMIT = -18
GM_ST_IN_USE = -22

GM_FT_INT64, GM_FT_DOUBLE, GM_FT_STRING = 1, 2, 3
GM_VS_OK, GM_VS_BLANK, GM_VS_NOT_SUPPORTED, GM_VS_NO_PERMISSION, GM_VS_ERROR = 0, 1, 2, 3, 4

GM_MODE_EMBEDDED, GM_MODE_STANDALONE, GM_MODE_START_HOSTENGINE = 0, 1, 2

# field ids used from Python (the full registry is queried at run time)
FI = dict(
    DRIVER_VERSION=1, DEV_NAME=50, DEV_UUID=54, DEV_PCI_BDF=57, GFX_CLOCK=100, MEM_CLOCK=101,
    MEM_TEMP=140, GPU_TEMP=150, POWER_USAGE=155, TOTAL_ENERGY=156, PCIE_TX_KB=200, PCIE_RX_KB=201,
    PCIE_REPLAY_COUNTER=202, GFX_UTIL=203, MEM_UTIL=204, ENC_UTIL=206, DEC_UTIL=207,
    LAST_EVENT_CODE=230, POWER_VIOLATION=240, THERMAL_VIOLATION=241, VRAM_TOTAL=250, VRAM_FREE=251,
    VRAM_USED=252, ECC_CE_VOLATILE=310, ECC_UE_VOLATILE=311, ECC_UMC_UE=321, BAD_PAGES_CE=390,
    BAD_PAGES_UE=391, BAD_PAGES_PENDING=392, XGMI_CRC_ERRORS=409, XGMI_DATA_ERRORS=419,
    XGMI_BANDWIDTH_TOTAL=449, XGMI_LINK_STATUS_0=470, XGMI_LINKS_DOWN=481, XGMI_ERROR_STATUS=482,
    PROF_GR_ENGINE_ACTIVE=1001, PROF_CU_ACTIVE=1002, PROF_MFMA_ACTIVE=1004, PROF_MEM_PIPE_ACTIVE=1020, LOW_UTIL_VIOLATION=244, RELIABILITY_VIOLATION=245, THROTTLE_STATUS=246, GFX_BELOW_HOST_LIMIT=247, ECC_DEFERRED=314,
)
THROTTLE = dict(POWER=1, THERMAL=2, HBM_THERMAL=4, VR_THERMAL=8, PROCHOT=16, LOW_UTIL=32, HOST_LIMIT=64)

HEALTH = dict(PCIE=1, XGMI=2, MEM=4, COMPUTE=8, THERMAL=16, POWER=32, DRIVER=64, FIRMWARE=128, ALL=0xFF)
HEALTH_PASS, HEALTH_WARN, HEALTH_FAIL = 0, 10, 20
POLICY = dict(ECC_UE=1, PCIE=2, MAX_BAD_PAGES=4, THERMAL=8, POWER=16, XGMI=32, GPU_EVENT=64, ALL=0x7F)
EVENT = dict(NONE=0, VMFAULT=1, THERMAL_THROTTLE=2, GPU_PRE_RESET=3, GPU_POST_RESET=4, RAS_UE=5,
             RAS_POISON=6, RING_HANG=7, DIAG_FAIL=8, PROCESS_START=20, PROCESS_END=21)


def _version(struct, n=1):
    return C.sizeof(struct) | (n << 24)


class _ValueUnion(C.Union):
    _fields_ = [("i64", C.c_int64), ("dbl", C.c_double), ("str", C.c_char * GM_STR_LEN)]


class Value(C.Structure):
    _fields_ = [("field_id", C.c_uint16), ("type", C.c_uint8), ("status", C.c_uint8), ("gpu", C.c_uint32),
                ("ts_us", C.c_int64), ("v", _ValueUnion)]

    def py(self):
        """Python value, or None when blank."""
        if self.status != GM_VS_OK:
            return None
        if self.type == GM_FT_DOUBLE:
            return self.v.dbl
        if self.type == GM_FT_STRING:
            return self.v.str.decode()
        return self.v.i64


class FieldMeta(C.Structure):
    _fields_ = [("field_id", C.c_uint16), ("type", C.c_uint8), ("is_static", C.c_uint8), ("prom_type", C.c_uint8),
                ("prom_class", C.c_uint8), ("reserved", C.c_uint16), ("tag", C.c_char_p), ("unit", C.c_char_p),
                ("prom_name", C.c_char_p), ("prom_help", C.c_char_p)]


class InitParams(C.Structure):
    _fields_ = [("version", C.c_uint32), ("mode", C.c_int32), ("address", C.c_char_p),
                ("address_is_unix_socket", C.c_int32), ("backend", C.c_char_p), ("backend_spec", C.c_char_p),
                ("hostengine_path", C.c_char_p)]


class DeviceAttributes(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("supported", C.c_int32),
                ("uuid", C.c_char * GM_STR_LEN), ("name", C.c_char * GM_STR_LEN), ("brand", C.c_char * GM_STR_LEN),
                ("serial", C.c_char * GM_STR_LEN), ("vbios", C.c_char * GM_STR_LEN),
                ("firmware", C.c_char * GM_STR_LEN), ("driver_version", C.c_char * GM_STR_LEN),
                ("pci_bdf", C.c_char * GM_STR_LEN), ("gfx_target", C.c_char * 32), ("device_path", C.c_char * 64),
                ("cpu_affinity", C.c_char * GM_STR_LEN), ("compute_partition", C.c_char * 16),
                ("memory_partition", C.c_char * 16), ("numa_node", C.c_int32), ("hip_id", C.c_int32),
                ("drm_render_minor", C.c_int32), ("num_xcd", C.c_uint32), ("num_cu", C.c_uint32),
                ("pcie_max_gen", C.c_uint32), ("pcie_max_width", C.c_uint32), ("pcie_bandwidth_mbs", C.c_int64),
                ("power_cap_w", C.c_double), ("power_cap_default_w", C.c_double), ("vram_total_mib", C.c_int64),
                ("vis_vram_total_mib", C.c_int64), ("gfx_clock_max_mhz", C.c_int64), ("mem_clock_max_mhz", C.c_int64)]


class PeerLink(C.Structure):
    _fields_ = [("gpu", C.c_uint32), ("link", C.c_int32), ("hops", C.c_uint32), ("num_links", C.c_uint32),
                ("weight", C.c_uint64), ("min_bw_mbs", C.c_int64), ("max_bw_mbs", C.c_int64),
                ("p2p_accessible", C.c_int32), ("pci_bdf", C.c_char * GM_STR_LEN)]


class XgmiLink(C.Structure):
    _fields_ = [("link", C.c_uint32), ("state", C.c_int32), ("peer_gpu", C.c_int32), ("peer_bdf", C.c_char * GM_STR_LEN),
                ("bit_rate_gbps", C.c_int64), ("max_bandwidth_gbps", C.c_int64), ("rx_kb", C.c_int64), ("tx_kb", C.c_int64)]


class HealthThresholds(C.Structure):
    _fields_ = [("version", C.c_uint32), ("hotspot_warn_c", C.c_int32), ("hotspot_fail_c", C.c_int32),
                ("mem_warn_c", C.c_int32), ("mem_fail_c", C.c_int32), ("bad_pages_fail", C.c_int32),
                ("ce_burst_warn", C.c_int32), ("power_over_cap_warn", C.c_double)]


class DiagResult(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("test", C.c_char * 32), ("system", C.c_uint32),
                ("health", C.c_int32), ("value", C.c_double), ("ts_us", C.c_int64), ("message", C.c_char * 192)]


class Topology(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("cpu_affinity", C.c_char * GM_STR_LEN),
                ("numa_node", C.c_int32), ("num_peers", C.c_uint32), ("peers", PeerLink * GM_MAX_GPUS)]


class DeviceStatus(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("valid", C.c_uint64), ("power_w", C.c_double),
                ("temperature_c", C.c_int64), ("gfx_util", C.c_int64), ("mem_util", C.c_int64),
                ("enc_util", C.c_int64), ("dec_util", C.c_int64), ("gfx_clock_mhz", C.c_int64),
                ("mem_clock_mhz", C.c_int64), ("vis_vram_used_mib", C.c_int64), ("pcie_rx_kb", C.c_int64),
                ("pcie_tx_kb", C.c_int64), ("pcie_replays", C.c_int64), ("vram_used_mib", C.c_int64),
                ("vram_free_mib", C.c_int64), ("ecc_ce", C.c_int64), ("ecc_ue", C.c_int64),
                ("perf_level", C.c_int64), ("fan_speed", C.c_int64), ("throttle_status", C.c_int64),
                ("mem_temp_c", C.c_int64), ("energy_mj", C.c_int64)]

    ORDER = ["power_w", "temperature_c", "gfx_util", "mem_util", "enc_util", "dec_util", "gfx_clock_mhz",
             "mem_clock_mhz", "vis_vram_used_mib", "pcie_rx_kb", "pcie_tx_kb", "pcie_replays", "vram_used_mib",
             "vram_free_mib", "ecc_ce", "ecc_ue", "perf_level", "fan_speed", "throttle_status", "mem_temp_c",
             "energy_mj"]

    def as_dict(self):
        return {n: (getattr(self, n) if (self.valid >> i) & 1 else None) for i, n in enumerate(self.ORDER)}


class Process(C.Structure):
    _fields_ = [("pid", C.c_uint32), ("type", C.c_uint32), ("vram_bytes", C.c_uint64), ("gtt_bytes", C.c_uint64),
                ("gfx_engine_ns", C.c_uint64), ("cu_occupancy", C.c_uint32), ("name", C.c_char * GM_STR_LEN)]


class DeviceMode(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("accounting_enabled", C.c_int32),
                ("accounting_buffer_size", C.c_int32), ("compute_partition", C.c_char * 16),
                ("memory_partition", C.c_char * 16)]


class Incident(C.Structure):
    _fields_ = [("system", C.c_uint32), ("health", C.c_int32), ("message", C.c_char * 192)]


class HealthResponse(C.Structure):
    _fields_ = [("version", C.c_uint32), ("gpu", C.c_uint32), ("overall", C.c_int32),
                ("incident_count", C.c_uint32), ("incidents", Incident * GM_MAX_INCIDENTS)]


class Policy(C.Structure):
    _fields_ = [("version", C.c_uint32), ("conditions", C.c_uint32), ("max_bad_pages", C.c_uint32),
                ("thermal_c", C.c_uint32), ("power_w", C.c_uint32), ("hysteresis_pct", C.c_uint32)]


class _VEcc(C.Structure):
    _fields_ = [("block", C.c_uint32), ("count", C.c_uint32), ("location", C.c_char * 32)]


class _VPcie(C.Structure):
    _fields_ = [("replay_counter", C.c_uint32)]


class _VPages(C.Structure):
    _fields_ = [("ce_pages", C.c_uint32), ("ue_pages", C.c_uint32)]


class _VThermal(C.Structure):
    _fields_ = [("temperature_c", C.c_uint32)]


class _VPower(C.Structure):
    _fields_ = [("power_w", C.c_uint32)]


class _VXgmi(C.Structure):
    _fields_ = [("field_id", C.c_uint16), ("link", C.c_uint16), ("counter", C.c_uint32)]


class _VEvent(C.Structure):
    _fields_ = [("code", C.c_uint32), ("message", C.c_char * 64)]


class _ViolationData(C.Union):
    _fields_ = [("ecc", _VEcc), ("pcie", _VPcie), ("bad_pages", _VPages), ("thermal", _VThermal),
                ("power", _VPower), ("xgmi", _VXgmi), ("event", _VEvent)]


class Violation(C.Structure):
    _fields_ = [("condition", C.c_uint32), ("gpu", C.c_uint32), ("ts_us", C.c_int64), ("d", _ViolationData)]


VIOLATION_CB = C.CFUNCTYPE(C.c_int, C.POINTER(Violation), C.c_void_p)


class PidGpuStats(C.Structure):
    _fields_ = [("gpu", C.c_uint32), ("pid", C.c_uint32), ("valid", C.c_uint64), ("start_ts_us", C.c_int64),
                ("end_ts_us", C.c_int64), ("energy_mj", C.c_int64), ("gfx_util_avg", C.c_double),
                ("mem_util_avg", C.c_double), ("proc_gfx_util_avg", C.c_double), ("proc_mem_util_avg", C.c_double),
                ("pcie_rx_avg_kbs", C.c_int64), ("pcie_tx_avg_kbs", C.c_int64), ("pcie_replays", C.c_int64),
                ("max_vram_bytes", C.c_int64), ("ecc_ce", C.c_int64), ("ecc_ue", C.c_int64),
                ("gfx_clock_avg", C.c_int64), ("mem_clock_avg", C.c_int64), ("power_violation_us", C.c_int64),
                ("thermal_violation_us", C.c_int64), ("reliability_violation_us", C.c_int64),
                ("board_limit_violation_us", C.c_int64), ("low_util_violation_us", C.c_int64),
                ("sync_boost_violation_us", C.c_int64), ("num_critical_events", C.c_uint32),
                ("critical_event_ts_us", C.c_int64 * GM_MAX_EVENT_TS), ("name", C.c_char * GM_STR_LEN)]


class PidInfo(C.Structure):
    _fields_ = [("version", C.c_uint32), ("pid", C.c_uint32), ("num_gpus", C.c_uint32),
                ("gpus", PidGpuStats * GM_MAX_GPUS)]


class Introspect(C.Structure):
    _fields_ = [("version", C.c_uint32), ("memory_kb", C.c_int64), ("cpu_percent", C.c_double),
                ("uptime_us", C.c_int64), ("sample_cycles", C.c_uint64), ("backend_errors", C.c_uint64),
                ("last_cycle_ms", C.c_double), ("avg_cycle_ms", C.c_double)]


class Event(C.Structure):
    _fields_ = [("gpu", C.c_uint32), ("code", C.c_uint32), ("ts_us", C.c_int64), ("uuid", C.c_char * GM_STR_LEN),
                ("message", C.c_char * 96)]


class ExporterConfig(C.Structure):
    _fields_ = [("version", C.c_uint32), ("profiling", C.c_int32), ("amd_extras", C.c_int32),
                ("self_metrics", C.c_int32), ("interval_ms", C.c_int64), ("gpus", C.c_char_p),
                ("output_path", C.c_char_p), ("listen", C.c_char_p), ("kubelet_socket", C.c_char_p),
                ("resource_names", C.c_char_p), ("pod_map_path", C.c_char_p), ("rest_api", C.c_int32),
                ("legacy_layout", C.c_int32), ("diag_results_path", C.c_char_p), ("state_path", C.c_char_p),
                ("health", C.c_int32), ("slow_interval_ms", C.c_int32), ("pod_refresh_ms", C.c_int32)]
    VERSION = 5


class PodExporterConfig(C.Structure):
    _fields_ = [("version", C.c_uint32), ("source_path", C.c_char_p), ("dest_path", C.c_char_p),
                ("kubelet_socket", C.c_char_p), ("resource_names", C.c_char_p), ("index_prefix", C.c_char_p),
                ("listen", C.c_char_p), ("timeout_ms", C.c_int32), ("liveness_ms", C.c_int32),
                ("pod_refresh_ms", C.c_int32)]


def versioned(struct_type):
    s = struct_type()
    s.version = _version(struct_type, getattr(struct_type, "VERSION", 1))
    return s
