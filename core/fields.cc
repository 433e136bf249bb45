#include "fields.h"

#include <cstring>
#include <mutex>
#include <unordered_map>

namespace gpumon {
namespace {

#define I64 GM_FT_INT64
#define DBL GM_FT_DOUBLE
#define STR GM_FT_STRING
#define GAUGE 1
#define COUNTER 2

// clang-format off
std::vector<FieldDef> build_table() {
  std::vector<FieldDef> t = {
    // ---- identity / static ------------------------------------------------
    {GM_FI_DRIVER_VERSION,        STR, SRC_STATIC, "driver_version", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_NAME,              STR, SRC_STATIC, "name", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_BRAND,             STR, SRC_STATIC, "brand", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_SERIAL,            STR, SRC_STATIC, "serial", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_UUID,              STR, SRC_STATIC, "uuid", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_PCI_BDF,           STR, SRC_STATIC, "pci_bdf", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_VBIOS_VERSION,     STR, SRC_STATIC, "vbios_version", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_FIRMWARE_VERSION,  STR, SRC_STATIC, "firmware_version", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_GFX_TARGET,        STR, SRC_STATIC, "gfx_target", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_DRM_RENDER_MINOR,  I64, SRC_STATIC, "drm_render_minor", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_NUMA_NODE,         I64, SRC_STATIC, "numa_node", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_CPU_AFFINITY,      STR, SRC_STATIC, "cpu_affinity", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_NUM_XCD,           I64, SRC_STATIC, "num_xcd", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_NUM_CU,            I64, SRC_STATIC, "num_cu", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_COMPUTE_PARTITION, STR, SRC_STATIC, "compute_partition", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_MEMORY_PARTITION,  STR, SRC_STATIC, "memory_partition", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_HIP_ID,            I64, SRC_STATIC, "hip_id", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_OAM_ID,            I64, SRC_STATIC, "oam_id", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_PARTITION_ID,      I64, SRC_STATIC, "partition_id", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_PCIE_MAX_GEN,      I64, SRC_STATIC, "pcie_max_gen", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_PCIE_MAX_WIDTH,    I64, SRC_STATIC, "pcie_max_width", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_POWER_CAP_DEFAULT, DBL, SRC_STATIC, "power_cap_default", "W", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_POWER_CAP_MAX,     DBL, SRC_STATIC, "power_cap_max", "W", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_GFX_CLOCK_MAX,     I64, SRC_STATIC, "gfx_clock_max", "MHz", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_MEM_CLOCK_MAX,     I64, SRC_STATIC, "mem_clock_max", "MHz", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_VIS_VRAM_TOTAL,    I64, SRC_STATIC, "vis_vram_total", "MiB", 0, 0, nullptr, nullptr, nullptr, 0},

    // ---- clocks -----------------------------------------------------------
    {GM_FI_GFX_CLOCK, I64, SRC_METRICS, "gfx_clock", "MHz", GAUGE, 0, "dcgm_sm_clock", "SM clock frequency (in MHz).", nullptr, 0},
    {GM_FI_MEM_CLOCK, I64, SRC_METRICS, "mem_clock", "MHz", GAUGE, 0, "dcgm_memory_clock", "Memory clock frequency (in MHz).", nullptr, 0},
    {GM_FI_SOC_CLOCK, I64, SRC_METRICS, "soc_clock", "MHz", GAUGE, 2, "gpumon_soc_clock", "SoC fabric clock frequency (in MHz).", nullptr, 0},
    {GM_FI_VCLK,      I64, SRC_METRICS, "vclk", "MHz", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DCLK,      I64, SRC_METRICS, "dclk", "MHz", 0, 0, nullptr, nullptr, nullptr, 0},

    // ---- temperatures -----------------------------------------------------
    {GM_FI_MEM_TEMP,    I64, SRC_METRICS, "mem_temp", "C", GAUGE, 0, "dcgm_memory_temp", "Memory temperature (in C).", nullptr, 0},
    {GM_FI_GPU_TEMP,    I64, SRC_METRICS, "gpu_temp", "C", GAUGE, 0, "dcgm_gpu_temp", "GPU temperature (in C).", nullptr, 0},
    {GM_FI_EDGE_TEMP,   I64, SRC_METRICS, "edge_temp", "C", GAUGE, 2, "gpumon_edge_temp", "Edge temperature (in C).", nullptr, 0},
    {GM_FI_VR_GFX_TEMP, I64, SRC_METRICS, "vr_gfx_temp", "C", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_VR_SOC_TEMP, I64, SRC_METRICS, "vr_soc_temp", "C", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_VR_MEM_TEMP, I64, SRC_METRICS, "vr_mem_temp", "C", 0, 0, nullptr, nullptr, nullptr, 0},

    // ---- power / energy ---------------------------------------------------
    {GM_FI_POWER_USAGE,  DBL, SRC_METRICS, "power_usage", "W", GAUGE, 0, "dcgm_power_usage", "Power draw (in W).", nullptr, 0},
    {GM_FI_TOTAL_ENERGY, I64, SRC_ENERGY,  "total_energy", "mJ", COUNTER, 0, "dcgm_total_energy_consumption", "Total energy consumption since boot (in mJ).", nullptr, 0},
    {GM_FI_DEV_POWER_CAP, DBL, SRC_PERF,   "power_cap", "W", GAUGE, 2, "gpumon_power_cap", "Enforced socket power cap (in W).", nullptr, 0},

    // ---- PCIe -------------------------------------------------------------
    {GM_FI_PCIE_TX_KB,          I64, SRC_METRICS, "pcie_tx", "KB", COUNTER, 0, "dcgm_pcie_tx_throughput", "Total number of bytes transmitted through PCIe TX (in KB) via NVML.", nullptr, 0},
    {GM_FI_PCIE_RX_KB,          I64, SRC_METRICS, "pcie_rx", "KB", COUNTER, 0, "dcgm_pcie_rx_throughput", "Total number of bytes received through PCIe RX (in KB) via NVML.", nullptr, 0},
    {GM_FI_PCIE_REPLAY_COUNTER, I64, SRC_METRICS, "pcie_replay", "", COUNTER, 0, "dcgm_pcie_replay_counter", "Total number of PCIe retries.", nullptr, 0},
    {GM_FI_PCIE_BW_INST,        I64, SRC_METRICS, "pcie_bw_inst", "MB/s", GAUGE, 2, "gpumon_pcie_bandwidth", "Instantaneous PCIe bandwidth (in MB/s).", nullptr, 0},
    {GM_FI_PCIE_L0_TO_RECOVERY, I64, SRC_METRICS, "pcie_l0_to_recovery", "", COUNTER, 2, "gpumon_pcie_l0_to_recovery_total", "Total number of PCIe L0 to recovery transitions.", nullptr, 0},
    {GM_FI_PCIE_NAK_SENT,       I64, SRC_METRICS, "pcie_nak_sent", "", COUNTER, 2, "gpumon_pcie_nak_sent_total", "Total number of PCIe NAKs sent.", nullptr, 0},
    {GM_FI_PCIE_NAK_RCVD,       I64, SRC_METRICS, "pcie_nak_rcvd", "", COUNTER, 2, "gpumon_pcie_nak_received_total", "Total number of PCIe NAKs received.", nullptr, 0},
    {GM_FI_DEV_PCIE_GEN,        I64, SRC_METRICS, "pcie_gen", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_PCIE_WIDTH,      I64, SRC_METRICS, "pcie_width", "", 0, 0, nullptr, nullptr, nullptr, 0},

    // ---- utilisation ------------------------------------------------------
    {GM_FI_GFX_UTIL, I64, SRC_METRICS, "gfx_util", "%", GAUGE, 0, "dcgm_gpu_utilization", "GPU utilization (in %).", nullptr, 0},
    {GM_FI_MEM_UTIL, I64, SRC_METRICS, "mem_util", "%", GAUGE, 0, "dcgm_mem_copy_utilization", "Memory utilization (in %).", nullptr, 0},
    {GM_FI_ENC_UTIL, I64, SRC_METRICS, "enc_util", "%", GAUGE, 0, "dcgm_enc_utilization", "Encoder utilization (in %).", nullptr, 0},
    {GM_FI_DEC_UTIL, I64, SRC_METRICS, "dec_util", "%", GAUGE, 0, "dcgm_dec_utilization", "Decoder utilization (in %).", nullptr, 0},

    // ---- events / throttling ----------------------------------------------
    {GM_FI_LAST_EVENT_CODE,       I64, SRC_EVENTS,    "last_event_code", "", GAUGE, 0, "dcgm_xid_errors", "Value of the last XID error encountered.", nullptr, 0},
    {GM_FI_POWER_VIOLATION,       I64, SRC_VIOLATION, "power_violation", "us", COUNTER, 0, "dcgm_power_violation", "Throttling duration due to power constraints (in us).", nullptr, 0},
    {GM_FI_THERMAL_VIOLATION,     I64, SRC_VIOLATION, "thermal_violation", "us", COUNTER, 0, "dcgm_thermal_violation", "Throttling duration due to thermal constraints (in us).", nullptr, 0},
    {GM_FI_SYNC_BOOST_VIOLATION,  I64, SRC_VIOLATION, "sync_boost_violation", "us", COUNTER, 0, "dcgm_sync_boost_violation", "Throttling duration due to sync-boost constraints (in us).", nullptr, 0},
    {GM_FI_BOARD_LIMIT_VIOLATION, I64, SRC_VIOLATION, "board_limit_violation", "us", COUNTER, 0, "dcgm_board_limit_violation", "Throttling duration due to board limit constraints (in us).", nullptr, 0},
    {GM_FI_LOW_UTIL_VIOLATION,    I64, SRC_VIOLATION, "low_util_violation", "us", COUNTER, 0, "dcgm_low_util_violation", "Throttling duration due to low utilization (in us).", nullptr, 0},
    {GM_FI_RELIABILITY_VIOLATION, I64, SRC_VIOLATION, "reliability_violation", "us", COUNTER, 0, "dcgm_reliability_violation", "Throttling duration due to reliability constraints (in us).", nullptr, 0},
    {GM_FI_GFX_BELOW_HOST_LIMIT,  I64, SRC_VIOLATION, "gfx_below_host_limit", "us", COUNTER, 2, "gpumon_gfx_clock_below_host_limit_us", "Time the shader clock has spent below the limit set by the host, for whatever reason, idling included (in us).", nullptr, 0},
    {GM_FI_THROTTLE_STATUS,       I64, SRC_VIOLATION, "throttle_status", "", GAUGE, 2, "gpumon_throttle_status", "Active clock throttle reasons (bitset: 1 power, 2 thermal, 4 HBM thermal, 8 VR thermal, 16 PROCHOT, 32 low utilization, 64 host limit).", nullptr, 0},
    {GM_FI_PERF_LEVEL,            I64, SRC_PERF,      "perf_level", "", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_FAN_SPEED,             I64, SRC_METRICS,   "fan_speed", "%", 0, 0, nullptr, nullptr, nullptr, 0},

    // ---- HBM capacity -----------------------------------------------------
    // The fb_total HELP text repeats "free": that typo is part of the text contract
    // (exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:156).
    {GM_FI_VRAM_TOTAL, I64, SRC_VRAM, "vram_total", "MiB", GAUGE, 0, "dcgm_fb_total", "Framebuffer memory free (in MiB).", nullptr, 0},
    {GM_FI_VRAM_FREE,  I64, SRC_VRAM, "vram_free", "MiB", GAUGE, 0, "dcgm_fb_free", "Framebuffer memory free (in MiB).", nullptr, 0},
    {GM_FI_VRAM_USED,  I64, SRC_VRAM, "vram_used", "MiB", GAUGE, 0, "dcgm_fb_used", "Framebuffer memory used (in MiB).", nullptr, 0},
    {GM_FI_DEV_VIS_VRAM_USED, I64, SRC_VRAM, "vis_vram_used", "MiB", 0, 0, nullptr, nullptr, nullptr, 0},
    {GM_FI_DEV_GTT_USED,      I64, SRC_VRAM, "gtt_used", "MiB", GAUGE, 2, "gpumon_gtt_used", "GTT (host memory mapped for the GPU) used (in MiB).", nullptr, 0},

    // ---- RAS --------------------------------------------------------------
    {GM_FI_ECC_CE_VOLATILE,  I64, SRC_ECC, "ecc_ce_volatile", "", COUNTER, 0, "dcgm_ecc_sbe_volatile_total", "Total number of single-bit volatile ECC errors.", nullptr, 0},
    {GM_FI_ECC_UE_VOLATILE,  I64, SRC_ECC, "ecc_ue_volatile", "", COUNTER, 0, "dcgm_ecc_dbe_volatile_total", "Total number of double-bit volatile ECC errors.", nullptr, 0},
    {GM_FI_ECC_CE_AGGREGATE, I64, SRC_ECC, "ecc_ce_aggregate", "", COUNTER, 0, "dcgm_ecc_sbe_aggregate_total", "Total number of single-bit persistent ECC errors.", nullptr, 0},
    {GM_FI_ECC_UE_AGGREGATE, I64, SRC_ECC, "ecc_ue_aggregate", "", COUNTER, 0, "dcgm_ecc_dbe_aggregate_total", "Total number of double-bit persistent ECC errors.", nullptr, 0},
    {GM_FI_ECC_DEFERRED,     I64, SRC_ECC, "ecc_deferred", "", COUNTER, 2, "gpumon_ecc_deferred_total", "Total number of deferred (poison) RAS errors.", nullptr, 0},
    {GM_FI_ECC_UMC_CE,   I64, SRC_ECC, "ecc_umc_ce", "", COUNTER, 2, "gpumon_ecc_block_correctable_total", "Total number of correctable RAS errors per block.", "block", 0},
    {GM_FI_ECC_UMC_UE,   I64, SRC_ECC, "ecc_umc_ue", "", COUNTER, 2, "gpumon_ecc_block_uncorrectable_total", "Total number of uncorrectable RAS errors per block.", "block", 0},
    {GM_FI_ECC_GFX_CE,   I64, SRC_ECC, "ecc_gfx_ce", "", COUNTER, 2, "gpumon_ecc_block_correctable_total", "Total number of correctable RAS errors per block.", "block", 1},
    {GM_FI_ECC_GFX_UE,   I64, SRC_ECC, "ecc_gfx_ue", "", COUNTER, 2, "gpumon_ecc_block_uncorrectable_total", "Total number of uncorrectable RAS errors per block.", "block", 1},
    {GM_FI_ECC_SDMA_CE,  I64, SRC_ECC, "ecc_sdma_ce", "", COUNTER, 2, "gpumon_ecc_block_correctable_total", "Total number of correctable RAS errors per block.", "block", 2},
    {GM_FI_ECC_SDMA_UE,  I64, SRC_ECC, "ecc_sdma_ue", "", COUNTER, 2, "gpumon_ecc_block_uncorrectable_total", "Total number of uncorrectable RAS errors per block.", "block", 2},
    {GM_FI_ECC_MMHUB_CE, I64, SRC_ECC, "ecc_mmhub_ce", "", COUNTER, 2, "gpumon_ecc_block_correctable_total", "Total number of correctable RAS errors per block.", "block", 3},
    {GM_FI_ECC_MMHUB_UE, I64, SRC_ECC, "ecc_mmhub_ue", "", COUNTER, 2, "gpumon_ecc_block_uncorrectable_total", "Total number of uncorrectable RAS errors per block.", "block", 3},
    {GM_FI_ECC_XGMI_CE,  I64, SRC_ECC, "ecc_xgmi_ce", "", COUNTER, 2, "gpumon_ecc_block_correctable_total", "Total number of correctable RAS errors per block.", "block", 4},
    {GM_FI_ECC_XGMI_UE,  I64, SRC_ECC, "ecc_xgmi_ue", "", COUNTER, 2, "gpumon_ecc_block_uncorrectable_total", "Total number of uncorrectable RAS errors per block.", "block", 4},

    // ---- bad pages --------------------------------------------------------
    {GM_FI_BAD_PAGES_CE,      I64, SRC_BAD_PAGES, "bad_pages_ce", "", COUNTER, 0, "dcgm_retired_pages_sbe", "Total number of retired pages due to single-bit errors.", nullptr, 0},
    {GM_FI_BAD_PAGES_UE,      I64, SRC_BAD_PAGES, "bad_pages_ue", "", COUNTER, 0, "dcgm_retired_pages_dbe", "Total number of retired pages due to double-bit errors.", nullptr, 0},
    {GM_FI_BAD_PAGES_PENDING, I64, SRC_BAD_PAGES, "bad_pages_pending", "", COUNTER, 0, "dcgm_retired_pages_pending", "Total number of pages pending retirement.", nullptr, 0},

    // ---- xGMI -------------------------------------------------------------
    {GM_FI_XGMI_CRC_ERRORS,      I64, SRC_ECC,     "xgmi_crc_errors", "", COUNTER, 0, "dcgm_nvlink_flit_crc_error_count_total", "Total number of NVLink flow-control CRC errors.", nullptr, 0},
    {GM_FI_XGMI_DATA_ERRORS,     I64, SRC_ECC,     "xgmi_data_errors", "", COUNTER, 0, "dcgm_nvlink_data_crc_error_count_total", "Total number of NVLink data CRC errors.", nullptr, 0},
    {GM_FI_XGMI_REPLAY_ERRORS,   I64, SRC_ECC,     "xgmi_replay_errors", "", COUNTER, 0, "dcgm_nvlink_replay_error_count_total", "Total number of NVLink retries.", nullptr, 0},
    {GM_FI_XGMI_RECOVERY_ERRORS, I64, SRC_ECC,     "xgmi_recovery_errors", "", COUNTER, 0, "dcgm_nvlink_recovery_error_count_total", "Total number of NVLink recovery errors.", nullptr, 0},
    {GM_FI_XGMI_BANDWIDTH_TOTAL, I64, SRC_METRICS, "xgmi_bandwidth_total", "KB", COUNTER, 0, "dcgm_nvlink_bandwidth_total", "Total number of NVLink bandwidth counters for all lanes", nullptr, 0},
    {GM_FI_XGMI_LINK_WIDTH,      I64, SRC_METRICS, "xgmi_link_width", "", GAUGE, 2, "gpumon_xgmi_link_width", "xGMI link width (lanes).", nullptr, 0},
    {GM_FI_XGMI_LINK_SPEED,      I64, SRC_METRICS, "xgmi_link_speed", "Gbps", GAUGE, 2, "gpumon_xgmi_link_speed", "xGMI link speed (in Gbps per lane).", nullptr, 0},
    {GM_FI_XGMI_LINKS_UP,        I64, SRC_METRICS, "xgmi_links_up", "", GAUGE, 2, "gpumon_xgmi_links_up", "Number of xGMI links that are up.", nullptr, 0},
    {GM_FI_XGMI_LINKS_DOWN,      I64, SRC_METRICS, "xgmi_links_down", "", GAUGE, 2, "gpumon_xgmi_links_down", "Number of xGMI links that are down.", nullptr, 0},
    {GM_FI_XGMI_ERROR_STATUS,    I64, SRC_ECC,     "xgmi_error_status", "", GAUGE, 2, "gpumon_xgmi_error_status", "xGMI link-layer error state (0 = no errors, 1 = an error, 2 = several errors).", nullptr, 0},

    // ---- profiling-class --------------------------------------------------
    {GM_FI_PROF_GR_ENGINE_ACTIVE, DBL, SRC_PROF, "prof_gr_engine_active", "%", GAUGE, 1, "dcgm_fi_prof_gr_engine_active", "Ratio of time the graphics engine is active (in %).", nullptr, 0},
    {GM_FI_PROF_CU_ACTIVE,        DBL, SRC_PROF, "prof_cu_active", "%", GAUGE, 1, "dcgm_fi_prof_sm_active", "The ratio of cycles an SM has at least 1 warp assigned (in %).", nullptr, 0},
    {GM_FI_PROF_CU_OCCUPANCY,     DBL, SRC_PROF, "prof_cu_occupancy", "%", GAUGE, 1, "dcgm_fi_prof_sm_occupancy", "The ratio of number of warps resident on an SM (in %).", nullptr, 0},
    {GM_FI_PROF_MFMA_ACTIVE,      DBL, SRC_PROF, "prof_mfma_active", "%", GAUGE, 1, "dcgm_fi_prof_pipe_tensor_active", "Ratio of cycles the tensor (HMMA) pipe is active (in %).", nullptr, 0},
    {GM_FI_PROF_MEM_PIPE_ACTIVE,  DBL, SRC_PROF, "prof_mem_pipe_active", "%", GAUGE, 2, "gpumon_prof_mem_pipe_active", "Ratio of cycles the vector-memory address units have work in flight (in %).", nullptr, 0},
    {GM_FI_PROF_HBM_ACTIVE,       DBL, SRC_PROF, "prof_hbm_active", "%", GAUGE, 1, "dcgm_fi_prof_dram_active", "Ratio of cycles the device memory interface is active sending or receiving data (in %).", nullptr, 0},
  };

  static char tagbuf[64][32];
  int tb = 0;
  auto tag = [&](const char* fmt, int i) {
    snprintf(tagbuf[tb], sizeof tagbuf[tb], fmt, i);
    return (const char*)tagbuf[tb++];
  };
  for (int i = 0; i < GM_MAX_HBM_STACKS; i++)
    t.push_back({(uint16_t)(GM_FI_HBM_STACK_TEMP_0 + i), I64, SRC_METRICS, tag("hbm_stack_temp_%d", i), "C", GAUGE, 2,
                 "gpumon_hbm_stack_temp", "HBM stack temperature (in C).", "stack", i});
  for (int i = 0; i < GM_MAX_XGMI_LINKS; i++) {
    t.push_back({(uint16_t)(GM_FI_XGMI_LINK_RX_KB_0 + i), I64, SRC_METRICS, tag("xgmi_link_rx_%d", i), "KB", COUNTER, 2,
                 "gpumon_xgmi_link_rx_kb_total", "Total data read over one xGMI link (in KB).", "link", i});
    t.push_back({(uint16_t)(GM_FI_XGMI_LINK_TX_KB_0 + i), I64, SRC_METRICS, tag("xgmi_link_tx_%d", i), "KB", COUNTER, 2,
                 "gpumon_xgmi_link_tx_kb_total", "Total data written over one xGMI link (in KB).", "link", i});
    t.push_back({(uint16_t)(GM_FI_XGMI_LINK_STATUS_0 + i), I64, SRC_METRICS, tag("xgmi_link_status_%d", i), "", GAUGE, 2,
                 "gpumon_xgmi_link_up", "xGMI link state (1 = up, 0 = down).", "link", i});
  }
  for (int i = 0; i < GM_MAX_XCD; i++)
    t.push_back({(uint16_t)(GM_FI_XCD_BUSY_0 + i), I64, SRC_METRICS, tag("xcd_busy_%d", i), "%", GAUGE, 2,
                 "gpumon_xcd_busy", "Graphics engine busy per XCD (in %).", "xcd", i});
  return t;
}
// clang-format on

struct Registry {
  std::vector<FieldDef> table;
  int16_t slot_of[GM_FI_MAX_ID];
  std::unordered_map<std::string, int> by_tag;
  std::vector<uint16_t> contract, profiling, extras;

  Registry() : table(build_table()) {
    memset(slot_of, 0xff, sizeof slot_of);
    for (size_t i = 0; i < table.size(); i++) {
      slot_of[table[i].id] = (int16_t)i;
      by_tag[table[i].tag] = (int)i;
    }
    // Contract order = column order of the reference's awk program.
    static const uint16_t order[] = {
        GM_FI_GFX_CLOCK, GM_FI_MEM_CLOCK, GM_FI_MEM_TEMP, GM_FI_GPU_TEMP, GM_FI_POWER_USAGE,
        GM_FI_TOTAL_ENERGY, GM_FI_PCIE_TX_KB, GM_FI_PCIE_RX_KB, GM_FI_PCIE_REPLAY_COUNTER,
        GM_FI_GFX_UTIL, GM_FI_MEM_UTIL, GM_FI_ENC_UTIL, GM_FI_DEC_UTIL, GM_FI_LAST_EVENT_CODE,
        GM_FI_POWER_VIOLATION, GM_FI_THERMAL_VIOLATION, GM_FI_SYNC_BOOST_VIOLATION,
        GM_FI_BOARD_LIMIT_VIOLATION, GM_FI_LOW_UTIL_VIOLATION, GM_FI_RELIABILITY_VIOLATION,
        GM_FI_VRAM_TOTAL, GM_FI_VRAM_FREE, GM_FI_VRAM_USED, GM_FI_ECC_CE_VOLATILE,
        GM_FI_ECC_UE_VOLATILE, GM_FI_ECC_CE_AGGREGATE, GM_FI_ECC_UE_AGGREGATE, GM_FI_BAD_PAGES_CE,
        GM_FI_BAD_PAGES_UE, GM_FI_BAD_PAGES_PENDING, GM_FI_XGMI_CRC_ERRORS, GM_FI_XGMI_DATA_ERRORS,
        GM_FI_XGMI_REPLAY_ERRORS, GM_FI_XGMI_RECOVERY_ERRORS, GM_FI_XGMI_BANDWIDTH_TOTAL};
    contract.assign(order, order + sizeof order / sizeof order[0]);
    for (auto& f : table) {
      if (f.prom_type && f.prom_class == 1) profiling.push_back(f.id);
      if (f.prom_type && f.prom_class == 2) extras.push_back(f.id);
    }
  }
};

const Registry& reg() {
  static const Registry r;
  return r;
}

}  // namespace

int field_slot(uint16_t id) { return id < GM_FI_MAX_ID ? reg().slot_of[id] : -1; }
const FieldDef* field_def(uint16_t id) {
  int s = field_slot(id);
  return s < 0 ? nullptr : &reg().table[s];
}
const FieldDef& field_at(int slot) { return reg().table[slot]; }
int field_count() { return (int)reg().table.size(); }
const FieldDef* field_by_tag(const std::string& tag) {
  auto it = reg().by_tag.find(tag);
  return it == reg().by_tag.end() ? nullptr : &reg().table[it->second];
}
const std::vector<uint16_t>& contract_fields() { return reg().contract; }
const std::vector<uint16_t>& profiling_fields() { return reg().profiling; }
const std::vector<uint16_t>& extra_fields() { return reg().extras; }

}  // namespace gpumon
