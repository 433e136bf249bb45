// What a healthy GPU is expected to deliver in the active diagnostics, and the grading against it.
//
// The reference's health check has three outcomes, Healthy / Warning / Failure
// (bindings/go/dcgm/health.go:88-98); its inputs are passive counters.  Here the active tests feed the same three
// outcomes: a measurement passes at or above 90 % of its expectation, warns below that and fails below 75 %
// (limits such as temperatures warn at the first level and fail at the second).  Expectations come from a table
// keyed by the GPU's architecture and product name, every entry of which can be replaced on the command line
// (--expect key=value,...  or, through --config FILE, one site-wide file) and scaled as a whole (--expect-scale).
#pragma once

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace diag_expect {

struct Table {
  const char* arch;  // prefix of the HIP architecture string
  const char* name;  // substring of the product name; "" = any
  const char* label;
  // Throughputs: higher is better.
  double hbm_tbps[6];       // copy scale add triad read write, best iteration, >= 1 GiB buffers
  double memtest_write_tbps, memtest_verify_tbps, memtest_march_tbps;  // march: bytes read plus bytes written
  double mfma_tflops[4];    // bf16 fp8 f32 mxfp8, register-resident pipe peak
  double gemm_tflops[4];    // bf16 fp8 - mxfp8 at 8192^3, sustained, checks on
  // Host link: share of what the negotiated PCIe link carries in one direction (generation x width), per traffic
  // pattern: copy engines up, down, each direction with both busy; a kernel reading and writing host memory.
  double pcie_fraction[5];
  double soak_gfx_clock_mhz;     // average shader clock held under the GEMM soak
  double soak_power_fraction;    // average board power over the power cap: a part that cannot draw its power is held back by something
  // Limits: lower is better; {warn at, fail at}.
  double hotspot_c[2], hbm_c[2];
  double thermal_residency[2];   // fraction of the soak spent thermally throttled
};

// MI355X (gfx950, 256 CUs, 8 HBM3E stacks, 1400 W): what the boxes of this build's pool reach within +-3 %,
// rounded down.  Rates from `gpumon-diag hbm|memtest|mfma|gemm` at their default sizes.
inline const std::vector<Table>& tables() {
  static const std::vector<Table> t = {
      {"gfx950", "", "MI355X-class (gfx950)",
       {6.5, 6.5, 6.5, 6.5, 6.8, 7.0}, 6.8, 6.5, 6.4,
       {2200, 2300, 150, 4700},
       {1550, 3000, 0, 3000},
       {0.88, 0.88, 0.58, 0.82, 0.80},
       1700, 0.80,
       {95, 105}, {85, 95}, {0.01, 0.10}},
  };
  return t;
}

struct Expect {
  bool known = false;  // false: no table entry for this GPU, correctness is graded and performance only reported
  std::string label;
  std::map<std::string, double> v;  // flat key -> value, e.g. hbm.copy, gemm.bf16, soak.gfx_clock_mhz, limit.hotspot_c.warn
  double warn_ratio = 0.90, fail_ratio = 0.75;
  double scale = 1.0;  // multiplies every throughput expectation

  double get(const std::string& key) const {
    auto it = v.find(key);
    return it == v.end() ? 0 : it->second;
  }
  // Throughput expectation; 0 = none.  The scale is for rates (a stricter or laxer site, a seeded failure in a
  // test); what the soak must hold (clock, share of the power cap) and the limits are not rates and stay.
  double want(const std::string& key) const { return get(key) * (key.rfind("soak.", 0) == 0 || key.rfind("limit.", 0) == 0 ? 1.0 : scale); }
};

inline const char* const kStream[] = {"copy", "scale", "add", "triad", "read", "write"};
inline const char* const kDtype[] = {"bf16", "fp8", "f32", "mxfp8"};
inline const char* const kPcie[] = {"h2d", "d2h", "bidir", "kernel_read", "kernel_write"};

inline Expect lookup(const std::string& arch, const std::string& name) {
  Expect e;
  for (auto& t : tables()) {
    if (arch.rfind(t.arch, 0) != 0 || (t.name[0] && name.find(t.name) == std::string::npos)) continue;
    e.known = true;
    e.label = t.label;
    for (int i = 0; i < 6; i++) e.v[std::string("hbm.") + kStream[i]] = t.hbm_tbps[i];
    e.v["memtest.write"] = t.memtest_write_tbps;
    e.v["memtest.verify"] = t.memtest_verify_tbps;
    e.v["memtest.march"] = t.memtest_march_tbps;
    for (int i = 0; i < 4; i++) {
      e.v[std::string("mfma.") + kDtype[i]] = t.mfma_tflops[i];
      if (t.gemm_tflops[i] > 0) e.v[std::string("gemm.") + kDtype[i]] = t.gemm_tflops[i];
    }
    for (int i = 0; i < 5; i++) e.v[std::string("pcie.") + kPcie[i]] = t.pcie_fraction[i];
    e.v["soak.gfx_clock_mhz"] = t.soak_gfx_clock_mhz;
    e.v["soak.power_fraction"] = t.soak_power_fraction;
    e.v["limit.hotspot_c.warn"] = t.hotspot_c[0], e.v["limit.hotspot_c.fail"] = t.hotspot_c[1];
    e.v["limit.hbm_c.warn"] = t.hbm_c[0], e.v["limit.hbm_c.fail"] = t.hbm_c[1];
    e.v["limit.thermal_residency.warn"] = t.thermal_residency[0], e.v["limit.thermal_residency.fail"] = t.thermal_residency[1];
    break;
  }
  return e;
}

// "key=value,key=value": replaces or adds entries; also warn_ratio / fail_ratio.  False on a malformed item.
inline bool apply_overrides(const std::string& spec, Expect* e, std::string* bad_item) {
  size_t p = 0;
  while (p < spec.size()) {
    size_t c = spec.find(',', p);
    if (c == std::string::npos) c = spec.size();
    const std::string item = spec.substr(p, c - p);
    p = c + 1;
    if (item.empty()) continue;
    const size_t eq = item.find('=');
    char* end = nullptr;
    const double x = eq == std::string::npos ? 0 : strtod(item.c_str() + eq + 1, &end);
    if (eq == std::string::npos || eq == 0 || !end || *end || !std::isfinite(x) || x < 0) {
      if (bad_item) *bad_item = item;
      return false;
    }
    const std::string key = item.substr(0, eq);
    if (key == "warn_ratio") e->warn_ratio = x;
    else if (key == "fail_ratio") e->fail_ratio = x;
    else e->v[key] = x, e->known = true;
  }
  if (e->fail_ratio > e->warn_ratio) e->fail_ratio = e->warn_ratio;
  return true;
}

enum { PASS = 0, WARN = 10, FAIL = 20 };  // the engine's health scale (gm_health_t)

// Throughput against its expectation: 0 = no expectation, passes.
inline int grade_rate(double value, double want, const Expect& e) {
  if (!(want > 0)) return PASS;
  if (!std::isfinite(value) || value < e.fail_ratio * want) return FAIL;
  return value < e.warn_ratio * want ? WARN : PASS;
}
// A quantity with an upper limit.
inline int grade_limit(double value, double warn_at, double fail_at) {
  if (!std::isfinite(value)) return PASS;
  if (fail_at > 0 && value >= fail_at) return FAIL;
  return warn_at > 0 && value >= warn_at ? WARN : PASS;
}
inline const char* health_name(int h) { return h >= FAIL ? "FAIL" : h >= WARN ? "WARN" : "PASS"; }

}  // namespace diag_expect
