// A stand-in for libamd_smi.so, so that the real device backend (core/backend/amdsmi_backend.cc) runs in the CPU
// tests: enumeration, the bulk metrics table through the "library" and through the backend's own read of the
// driver's file, and what happens when the two disagree.  Only the entry points the backend requires are here
// (plus the ASIC info); every optional one is absent, which the backend must take as "not supported".
//
//   STUB_GPUS        number of GPUs, one per socket (default 2); GPU i sits at 0000:<10+i>:00.0
//   STUB_PARTITIONS  compute partitions per GPU (default 1): every socket then has that many processors, which share
//                    the PCI address and differ in partition id, handed out in REVERSE order so that the backend's
//                    sorting has something to do; the partition mode reads "CPX" (8), "QPX" (4), "DPX" (2) or "SPX"
//   STUB_SYSFS       tree that holds bus/pci/devices/<bdf>/gpu_metrics (format 1 content 8), which this "library"
//                    translates the way the real one does
//   STUB_SKEW        added to the hotspot temperature the library reports (a decoding that differs from the file)
//   STUB_CALLS_FILE  one byte appended per amdsmi_get_gpu_metrics_info call
//   STUB_SCRIPT      file of key=value lines read at every call of the optional entry points, so that a test can
//                    change the device under a running engine: ecc_ce ecc_ue ecc_deferred (totals), ecc_status
//                    (a library status code for the RAS calls: 0 ok, 10 = no permission, 2 = not supported),
//                    vram_used_mib, bad_reserved, bad_pending, perf_level, power_cap_w
// build: g++ -shared -fPIC -O1 -I/opt/rocm/include tests/stubs/amdsmi_stub.cc -o libamd_smi_stub.so
#include <amd_smi/amdsmi.h>

#include <unistd.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

namespace {

struct Gpu { int index, partition; } g_gpus[16 * 8];
struct Socket { int index; } g_sockets[16];
int g_n = 0, g_parts = 1;

int env_int(const char* k, int d) {
  const char* v = getenv(k);
  return v && *v ? atoi(v) : d;
}
template <typename T>
T rd(const unsigned char* raw, size_t off) {
  T v;
  memcpy(&v, raw + off, sizeof v);
  return v;
}
std::string bdf_of(int i) {
  char b[32];
  snprintf(b, sizeof b, "0000:%02x:00.0", 0x10 + i);
  return b;
}

}  // namespace

extern "C" {

amdsmi_status_t amdsmi_init(uint64_t) {
  g_n = env_int("STUB_GPUS", 2);
  if (g_n < 0 || g_n > 16) g_n = 2;
  g_parts = env_int("STUB_PARTITIONS", 1);
  if (g_parts < 1 || g_parts > 8) g_parts = 1;
  for (int i = 0; i < g_n; i++) {
    g_sockets[i].index = i;
    for (int p = 0; p < g_parts; p++) g_gpus[i * 8 + p] = Gpu{i, g_parts - 1 - p};
  }
  return AMDSMI_STATUS_SUCCESS;
}
// A call that is still inside the library when the library is shut down is the caller's fault; the real library does not
// say so, it goes on with freed state.  This one does: amdsmi_get_violation_status stays inside for STUB_VIOLATION_MS
// (the real one sleeps 100 ms between two reads of its own), and a shut-down that arrives meanwhile ends the process.
static std::atomic<int> g_inside{0};
amdsmi_status_t amdsmi_shut_down(void) {
  if (g_inside.load() != 0) {
    fprintf(stderr, "amdsmi stub: amdsmi_shut_down while %d call(s) are inside the library\n", g_inside.load());
    abort();
  }
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_violation_status(amdsmi_processor_handle, amdsmi_violation_status_t* v) {
  if (!v) return AMDSMI_STATUS_INVAL;
  const int ms = env_int("STUB_VIOLATION_MS", -1);
  if (ms < 0) return AMDSMI_STATUS_NOT_SUPPORTED;
  g_inside++;
  usleep((useconds_t)ms * 1000);
  memset(v, 0xff, sizeof *v);
  g_inside--;
  return AMDSMI_STATUS_SUCCESS;
}

amdsmi_status_t amdsmi_get_socket_handles(uint32_t* n, amdsmi_socket_handle* h) {
  if (!n) return AMDSMI_STATUS_INVAL;
  if (h)
    for (int i = 0; i < g_n && i < (int)*n; i++) h[i] = &g_sockets[i];
  *n = (uint32_t)g_n;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_processor_handles(amdsmi_socket_handle s, uint32_t* n, amdsmi_processor_handle* h) {
  if (!n || !s) return AMDSMI_STATUS_INVAL;
  if (h)
    for (int p = 0; p < g_parts && p < (int)*n; p++) h[p] = &g_gpus[static_cast<Socket*>(s)->index * 8 + p];
  *n = (uint32_t)g_parts;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_processor_type(amdsmi_processor_handle, processor_type_t* t) {
  *t = AMDSMI_PROCESSOR_TYPE_AMD_GPU;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_device_bdf(amdsmi_processor_handle h, amdsmi_bdf_t* bdf) {
  bdf->as_uint = 0;
  bdf->bus_number = 0x10 + static_cast<Gpu*>(h)->index;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_device_uuid(amdsmi_processor_handle h, unsigned int* len, char* uuid) {
  snprintf(uuid, *len, "stub-%04d-0000-1000-8000-0000%04d%04d", static_cast<Gpu*>(h)->index, static_cast<Gpu*>(h)->partition,
           static_cast<Gpu*>(h)->index);
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_kfd_info(amdsmi_processor_handle h, amdsmi_kfd_info_t* k) {
  memset(k, 0xff, sizeof *k);
  k->current_partition_id = (uint32_t)static_cast<Gpu*>(h)->partition;
  k->node_id = (uint32_t)(static_cast<Gpu*>(h)->index * 8 + static_cast<Gpu*>(h)->partition);
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_compute_partition(amdsmi_processor_handle, char* out, uint32_t len) {
  snprintf(out, len, "%s", g_parts == 8 ? "CPX" : g_parts == 4 ? "QPX" : g_parts == 2 ? "DPX" : "SPX");
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_asic_info(amdsmi_processor_handle, amdsmi_asic_info_t* a) {
  memset(a, 0, sizeof *a);
  snprintf(a->market_name, sizeof a->market_name, "AMD Instinct MI355X");
  a->device_id = 0x75a3;
  a->num_of_compute_units = 256;
  a->target_graphics_version = 0x950;
  return AMDSMI_STATUS_SUCCESS;
}

// ---- optional entry points, scripted --------------------------------------------------------------------------
static long script(const char* key, long dflt) {
  const char* path = getenv("STUB_SCRIPT");
  if (!path) return dflt;
  FILE* f = fopen(path, "r");
  if (!f) return dflt;
  char line[256];
  long v = dflt;
  const size_t kl = strlen(key);
  while (fgets(line, sizeof line, f))
    if (!strncmp(line, key, kl) && line[kl] == '=') v = atol(line + kl + 1);
  fclose(f);
  return v;
}

amdsmi_status_t amdsmi_get_gpu_memory_total(amdsmi_processor_handle, amdsmi_memory_type_t t, uint64_t* total) {
  *total = t == AMDSMI_MEM_TYPE_VRAM ? 294896ull << 20 : 1ull << 30;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_memory_usage(amdsmi_processor_handle, amdsmi_memory_type_t t, uint64_t* used) {
  if (t != AMDSMI_MEM_TYPE_VRAM) return AMDSMI_STATUS_NOT_SUPPORTED;
  *used = (uint64_t)script("vram_used_mib", 0) << 20;  // 0: what the real library says where the kernel counter is missing
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_total_ecc_count(amdsmi_processor_handle, amdsmi_error_count_t* ec) {
  const long st = script("ecc_status", 0);
  if (st) return (amdsmi_status_t)st;
  memset(ec, 0, sizeof *ec);
  ec->correctable_count = (uint64_t)script("ecc_ce", 0);
  ec->uncorrectable_count = (uint64_t)script("ecc_ue", 0);
  ec->deferred_count = (uint64_t)script("ecc_deferred", 0);
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_ecc_count(amdsmi_processor_handle, amdsmi_gpu_block_t block, amdsmi_error_count_t* ec) {
  const long st = script("ecc_status", 0);
  if (st) return (amdsmi_status_t)st;
  memset(ec, 0, sizeof *ec);
  if (block == AMDSMI_GPU_BLOCK_UMC) {  // all errors of the script are memory errors
    ec->correctable_count = (uint64_t)script("ecc_ce", 0);
    ec->uncorrectable_count = (uint64_t)script("ecc_ue", 0);
  }
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_bad_page_info(amdsmi_processor_handle, uint32_t* n, amdsmi_retired_page_record_t* recs) {
  const uint32_t reserved = (uint32_t)script("bad_reserved", 0), pending = (uint32_t)script("bad_pending", 0);
  if (recs)
    for (uint32_t i = 0; i < reserved + pending && i < *n; i++) {
      recs[i].page_address = 0x1000ull * (i + 1);
      recs[i].page_size = 4096;
      recs[i].status = i < reserved ? AMDSMI_MEM_PAGE_STATUS_RESERVED : AMDSMI_MEM_PAGE_STATUS_PENDING;
    }
  *n = reserved + pending;
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_gpu_perf_level(amdsmi_processor_handle, amdsmi_dev_perf_level_t* lvl) {
  *lvl = (amdsmi_dev_perf_level_t)script("perf_level", (long)AMDSMI_DEV_PERF_LEVEL_AUTO);
  return AMDSMI_STATUS_SUCCESS;
}
amdsmi_status_t amdsmi_get_power_cap_info(amdsmi_processor_handle, uint32_t, amdsmi_power_cap_info_t* p) {
  memset(p, 0, sizeof *p);
  p->power_cap = (uint64_t)script("power_cap_w", 1400) * 1000000ull;
  p->max_power_cap = 1400ull * 1000000ull;
  return AMDSMI_STATUS_SUCCESS;
}

// The translation of the 1.8 table into the all-versions struct, member by member, as the real library does it.
amdsmi_status_t amdsmi_get_gpu_metrics_info(amdsmi_processor_handle h, amdsmi_gpu_metrics_t* m) {
  if (const char* calls = getenv("STUB_CALLS_FILE")) {
    if (FILE* f = fopen(calls, "ab")) { fputc('.', f); fclose(f); }
  }
  const char* root = getenv("STUB_SYSFS");
  if (!root) return AMDSMI_STATUS_FILE_NOT_FOUND;
  const std::string path = std::string(root) + "/bus/pci/devices/" + bdf_of(static_cast<Gpu*>(h)->index) + "/gpu_metrics";
  unsigned char raw[4096];
  FILE* f = fopen(path.c_str(), "rb");
  if (!f) return AMDSMI_STATUS_FILE_NOT_FOUND;
  const size_t n = fread(raw, 1, sizeof raw, f);
  fclose(f);
  if (n < 3872 || raw[2] != 1 || raw[3] != 8) return AMDSMI_STATUS_NOT_SUPPORTED;
  memset(m, 0xff, sizeof *m);
  m->common_header.structure_size = rd<uint16_t>(raw, 0);
  m->common_header.format_revision = raw[2];
  m->common_header.content_revision = raw[3];
  m->temperature_hotspot = (uint16_t)(rd<uint16_t>(raw, 4) + env_int("STUB_SKEW", 0));
  m->temperature_mem = rd<uint16_t>(raw, 6);
  m->temperature_vrsoc = rd<uint16_t>(raw, 8);
  m->current_socket_power = rd<uint16_t>(raw, 10);
  m->average_gfx_activity = rd<uint16_t>(raw, 12);
  m->average_umc_activity = rd<uint16_t>(raw, 14);
  m->vram_max_bandwidth = rd<uint64_t>(raw, 16);
  m->energy_accumulator = rd<uint64_t>(raw, 24);
  m->system_clock_counter = rd<uint64_t>(raw, 32);
  m->accumulation_counter = rd<uint32_t>(raw, 40);
  m->prochot_residency_acc = rd<uint32_t>(raw, 44);
  m->ppt_residency_acc = rd<uint32_t>(raw, 48);
  m->socket_thm_residency_acc = rd<uint32_t>(raw, 52);
  m->vr_thm_residency_acc = rd<uint32_t>(raw, 56);
  m->hbm_thm_residency_acc = rd<uint32_t>(raw, 60);
  m->gfxclk_lock_status = rd<uint32_t>(raw, 64);
  m->pcie_link_width = rd<uint16_t>(raw, 68);
  m->pcie_link_speed = rd<uint16_t>(raw, 70);
  m->xgmi_link_width = rd<uint16_t>(raw, 72);
  m->xgmi_link_speed = rd<uint16_t>(raw, 74);
  m->gfx_activity_acc = rd<uint32_t>(raw, 76);
  m->mem_activity_acc = rd<uint32_t>(raw, 80);
  m->pcie_bandwidth_acc = rd<uint64_t>(raw, 88);
  m->pcie_bandwidth_inst = rd<uint64_t>(raw, 96);
  m->pcie_l0_to_recov_count_acc = rd<uint64_t>(raw, 104);
  m->pcie_replay_count_acc = rd<uint64_t>(raw, 112);
  m->pcie_replay_rover_count_acc = rd<uint64_t>(raw, 120);
  m->pcie_nak_sent_count_acc = rd<uint32_t>(raw, 128);
  m->pcie_nak_rcvd_count_acc = rd<uint32_t>(raw, 132);
  for (int i = 0; i < 8; i++) {
    m->xgmi_read_data_acc[i] = rd<uint64_t>(raw, 136 + 8 * i);
    m->xgmi_write_data_acc[i] = rd<uint64_t>(raw, 200 + 8 * i);
    m->xgmi_link_status[i] = rd<uint16_t>(raw, 264 + 2 * i);
    m->current_gfxclks[i] = rd<uint16_t>(raw, 296 + 2 * i);
  }
  m->firmware_timestamp = rd<uint64_t>(raw, 288);
  for (int i = 0; i < 4; i++) {
    m->current_socclks[i] = rd<uint16_t>(raw, 312 + 2 * i);
    m->current_vclk0s[i] = rd<uint16_t>(raw, 320 + 2 * i);
    m->current_dclk0s[i] = rd<uint16_t>(raw, 328 + 2 * i);
  }
  m->current_uclk = rd<uint16_t>(raw, 336);
  m->current_gfxclk = m->current_gfxclks[0];
  m->current_socclk = m->current_socclks[0];
  m->current_vclk0 = m->current_vclk0s[0];
  m->current_dclk0 = m->current_dclk0s[0];
  m->num_partition = rd<uint16_t>(raw, 338);
  for (int p = 0; p < 8; p++) {
    const size_t base = 344 + 440 * (size_t)p;
    for (int x = 0; x < 8; x++) {
      m->xcp_stats[p].gfx_busy_inst[x] = rd<uint32_t>(raw, base + 4 * x);
      m->xcp_stats[p].gfx_busy_acc[x] = rd<uint64_t>(raw, base + 120 + 8 * x);
      m->xcp_stats[p].gfx_below_host_limit_ppt_acc[x] = rd<uint64_t>(raw, base + 184 + 8 * x);
      m->xcp_stats[p].gfx_below_host_limit_thm_acc[x] = rd<uint64_t>(raw, base + 248 + 8 * x);
      m->xcp_stats[p].gfx_low_utilization_acc[x] = rd<uint64_t>(raw, base + 312 + 8 * x);
      m->xcp_stats[p].gfx_below_host_limit_total_acc[x] = rd<uint64_t>(raw, base + 376 + 8 * x);
    }
  }
  m->pcie_lc_perf_other_end_recovery = rd<uint32_t>(raw, 3864);
  return AMDSMI_STATUS_SUCCESS;
}

}  // extern "C"
