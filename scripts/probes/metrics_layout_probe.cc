// Prints the device library's decoded metrics table next to the raw sysfs table it was decoded from, for several
// moments, so that the raw layout can be matched field by field (scripts/metrics_layout_match.py).
// build: g++ -O1 -std=c++17 -I/opt/rocm/include scripts/probes/metrics_layout_probe.cc -o build/bin/metrics-layout-probe -L/opt/rocm/lib -lamd_smi -Wl,-rpath,/opt/rocm/lib
#include <amd_smi/amdsmi.h>
#include <fcntl.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

static std::string raw_read(int fd) {
  char buf[8192];
  ssize_t n = pread(fd, buf, sizeof buf, 0);
  return n > 0 ? std::string(buf, (size_t)n) : std::string();
}

#define F(name) printf("  \"" #name "\": %llu,\n", (unsigned long long)m.name)
#define A(name, n)                                                             \
  do {                                                                         \
    printf("  \"" #name "\": [");                                              \
    for (int i = 0; i < (n); i++) printf("%s%llu", i ? "," : "", (unsigned long long)m.name[i]); \
    printf("],\n");                                                            \
  } while (0)

int main(int argc, char** argv) {
  const char* path = argc > 1 ? argv[1] : "/sys/class/drm/card0/device/gpu_metrics";
  const int shots = argc > 2 ? atoi(argv[2]) : 4;
  if (amdsmi_init(AMDSMI_INIT_AMD_GPUS) != AMDSMI_STATUS_SUCCESS) return 1;
  uint32_t ns = 0;
  amdsmi_get_socket_handles(&ns, nullptr);
  std::vector<amdsmi_socket_handle> socks(ns);
  amdsmi_get_socket_handles(&ns, socks.data());
  amdsmi_processor_handle h = nullptr;
  for (auto s : socks) {
    uint32_t np = 0;
    amdsmi_get_processor_handles(s, &np, nullptr);
    std::vector<amdsmi_processor_handle> ps(np);
    amdsmi_get_processor_handles(s, &np, ps.data());
    if (np) { h = ps[0]; break; }
  }
  if (!h) return 2;
  int fd = open(path, O_RDONLY);
  if (fd < 0) return 3;
  printf("[\n");
  int done = 0;
  for (int attempt = 0; attempt < 400 && done < shots; attempt++) {
    amdsmi_gpu_metrics_t m;
    memset(&m, 0xff, sizeof m);
    std::string r0 = raw_read(fd);
    if (amdsmi_get_gpu_metrics_info(h, &m) != AMDSMI_STATUS_SUCCESS) return 4;
    std::string r1 = raw_read(fd);
    if (r0 != r1) continue;  // the table changed under the library's read: no clean pair
    printf("%s{\n  \"raw\": \"", done ? "," : "");
    for (unsigned char c : r0) printf("%02x", c);
    printf("\",\n");
    F(common_header.structure_size); F(common_header.format_revision); F(common_header.content_revision);
    F(temperature_edge); F(temperature_hotspot); F(temperature_mem); F(temperature_vrgfx); F(temperature_vrsoc); F(temperature_vrmem);
    F(average_gfx_activity); F(average_umc_activity); F(average_mm_activity); F(average_socket_power); F(energy_accumulator);
    F(system_clock_counter); F(average_gfxclk_frequency); F(average_socclk_frequency); F(average_uclk_frequency);
    F(current_gfxclk); F(current_socclk); F(current_uclk); F(current_vclk0); F(current_dclk0); F(throttle_status); F(current_fan_speed);
    F(pcie_link_width); F(pcie_link_speed); F(gfx_activity_acc); F(mem_activity_acc); A(temperature_hbm, AMDSMI_NUM_HBM_INSTANCES);
    F(firmware_timestamp); F(voltage_soc); F(voltage_gfx); F(voltage_mem); F(indep_throttle_status); F(current_socket_power);
    A(vcn_activity, AMDSMI_MAX_NUM_VCN); F(gfxclk_lock_status); F(xgmi_link_width); F(xgmi_link_speed); F(pcie_bandwidth_acc);
    F(pcie_bandwidth_inst); F(pcie_l0_to_recov_count_acc); F(pcie_replay_count_acc); F(pcie_replay_rover_count_acc);
    A(xgmi_read_data_acc, AMDSMI_MAX_NUM_XGMI_LINKS); A(xgmi_write_data_acc, AMDSMI_MAX_NUM_XGMI_LINKS);
    A(current_gfxclks, AMDSMI_MAX_NUM_GFX_CLKS); A(current_socclks, AMDSMI_MAX_NUM_CLKS); A(current_vclk0s, AMDSMI_MAX_NUM_CLKS);
    A(current_dclk0s, AMDSMI_MAX_NUM_CLKS); F(pcie_nak_sent_count_acc); F(pcie_nak_rcvd_count_acc); F(accumulation_counter);
    F(prochot_residency_acc); F(ppt_residency_acc); F(socket_thm_residency_acc); F(vr_thm_residency_acc); F(hbm_thm_residency_acc);
    F(num_partition); F(pcie_lc_perf_other_end_recovery); F(vram_max_bandwidth); A(xgmi_link_status, AMDSMI_MAX_NUM_XGMI_LINKS);
    for (int p = 0; p < AMDSMI_MAX_NUM_XCP; p++) {
      printf("  \"xcp%d\": {", p);
      auto& x = m.xcp_stats[p];
#define XA(name, n)                                                            \
  do {                                                                         \
    printf("\"" #name "\": [");                                                \
    for (int i = 0; i < (n); i++) printf("%s%llu", i ? "," : "", (unsigned long long)x.name[i]); \
    printf("], ");                                                             \
  } while (0)
      XA(gfx_busy_inst, AMDSMI_MAX_NUM_XCC); XA(vcn_busy, AMDSMI_MAX_NUM_VCN); XA(gfx_busy_acc, AMDSMI_MAX_NUM_XCC);
      XA(gfx_below_host_limit_acc, AMDSMI_MAX_NUM_XCC); XA(gfx_below_host_limit_ppt_acc, AMDSMI_MAX_NUM_XCC);
      XA(gfx_below_host_limit_thm_acc, AMDSMI_MAX_NUM_XCC); XA(gfx_low_utilization_acc, AMDSMI_MAX_NUM_XCC);
      XA(gfx_below_host_limit_total_acc, AMDSMI_MAX_NUM_XCC);
      printf("\"jpeg_busy0\": %llu},\n", (unsigned long long)x.jpeg_busy[0]);
    }
    printf("  \"sizeof\": %zu\n}\n", sizeof m);
    done++;
    usleep(700000);
  }
  printf("]\n");
  close(fd);
  amdsmi_shut_down();
  return done ? 0 : 5;
}
