// Policy module.  Keeps the observable contract of the reference's policy path
// — register conditions on GPUs, receive asynchronous violations that carry
// {condition, timestamp, condition-specific data} (bindings/go/dcgm/policy.go:
// 33-85, 162-249, 285-389) — and drops its defects: registrations stay live
// until unregistered (the reference closes after one event per condition),
// thresholds are configurable, and each threshold crossing fires exactly once
// with hysteresis before it re-arms.
//
// Conditions are evaluated on the sampler tick from the snapshot cache and on
// driver events; callbacks run on the notifier thread.
#include <cstring>

#include "../common/util.h"
#include "engine.h"

namespace gpumon {
namespace {

const uint16_t kPolicyFields[] = {GM_FI_ECC_UE_VOLATILE, GM_FI_ECC_UMC_UE, GM_FI_ECC_GFX_UE,
                                  GM_FI_PCIE_REPLAY_COUNTER, GM_FI_BAD_PAGES_CE, GM_FI_BAD_PAGES_UE,
                                  GM_FI_GPU_TEMP, GM_FI_POWER_USAGE, GM_FI_XGMI_CRC_ERRORS,
                                  GM_FI_XGMI_DATA_ERRORS, GM_FI_XGMI_LINKS_DOWN};

gm_policy_t default_policy() {
  gm_policy_t p;
  memset(&p, 0, sizeof p);
  p.version = GM_POLICY_VERSION;
  p.conditions = GM_POLICY_COND_ALL;
  p.max_bad_pages = 10;
  p.thermal_c = 100;
  p.power_w = 1400;
  p.hysteresis_pct = 3;
  return p;
}

}  // namespace

gm_status_t LocalEngine::policy_set(gm_group_t g, const gm_policy_t& p) {
  if (p.version != GM_POLICY_VERSION) return GM_ST_VER_MISMATCH;
  std::lock_guard<std::mutex> l(state_mu_);
  std::vector<uint32_t> gp;
  gm_status_t st = resolve_group(g, &gp);
  if (st != GM_ST_OK) return st;
  Group& grp = groups_[g];
  gm_policy_t cur = grp.has_policy ? grp.policy : default_policy();
  // Only the thresholds of the named conditions change.
  if (p.conditions & GM_POLICY_COND_MAX_BAD_PAGES) cur.max_bad_pages = p.max_bad_pages;
  if (p.conditions & GM_POLICY_COND_THERMAL) cur.thermal_c = p.thermal_c;
  if (p.conditions & GM_POLICY_COND_POWER) cur.power_w = p.power_w;
  if (p.hysteresis_pct) cur.hysteresis_pct = p.hysteresis_pct > 50 ? 50 : p.hysteresis_pct;
  cur.conditions |= p.conditions;
  grp.policy = cur;
  grp.has_policy = true;
  return GM_ST_OK;
}

gm_status_t LocalEngine::policy_get(gm_group_t g, gm_policy_t* p) {
  std::lock_guard<std::mutex> l(state_mu_);
  std::vector<uint32_t> gp;
  gm_status_t st = resolve_group(g, &gp);
  if (st != GM_ST_OK) return st;
  auto it = groups_.find(g);
  *p = (it != groups_.end() && it->second.has_policy) ? it->second.policy : default_policy();
  return GM_ST_OK;
}

gm_status_t LocalEngine::policy_register(gm_group_t g, uint32_t conditions, ViolationFn fn, gm_policy_reg_t* out) {
  if (!conditions || (conditions & ~GM_POLICY_COND_ALL) || !fn) return GM_ST_BADPARAM;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    std::vector<uint32_t> gp;
    gm_status_t st = resolve_group(g, &gp);
    if (st != GM_ST_OK) return st;
    if (gp.empty()) return GM_ST_GROUP_INCOMPATIBLE;
    auto reg = std::make_shared<PolicyReg>();
    reg->group = g;
    reg->conditions = conditions;
    reg->fn = std::move(fn);
    for (uint32_t gpu : gp) reg->arm[gpu] = PolicyArm();
    *out = next_reg_++;
    policy_regs_[*out] = reg;
    const size_t n = sizeof kPolicyFields / sizeof kPolicyFields[0];
    ensure_internal_watch("policy", gp, std::vector<uint16_t>(kPolicyFields, kPolicyFields + n),
                          GM_DEFAULT_UPDATE_FREQ_US);
  }
  // First sample establishes the counter baselines (an error that predates the
  // registration is not a violation of it).
  return update_all(true);
}

gm_status_t LocalEngine::policy_unregister(gm_policy_reg_t reg) {
  {
    std::lock_guard<std::mutex> l(state_mu_);
    if (!policy_regs_.erase(reg)) return GM_ST_NOT_FOUND;
  }
  notifier_.drain();
  return GM_ST_OK;
}

void LocalEngine::fire(const std::shared_ptr<PolicyReg>& reg, const gm_violation_t& v) {
  notifier_.post([reg, v] { reg->fn(v); });
}

void LocalEngine::evaluate_policies(int64_t now_wall) {
  struct Work {
    std::shared_ptr<PolicyReg> reg;
    gm_policy_t policy;
  };
  std::vector<Work> work;
  {
    std::lock_guard<std::mutex> l(state_mu_);
    if (policy_regs_.empty()) return;
    for (auto& kv : policy_regs_) {
      auto it = groups_.find(kv.second->group);
      gm_policy_t p = (it != groups_.end() && it->second.has_policy) ? it->second.policy : default_policy();
      work.push_back({kv.second, p});
    }
  }
  static std::mutex eval_mu;  // sampler tick and inject_value may race here
  std::lock_guard<std::mutex> el(eval_mu);
  for (auto& w : work) {
    PolicyReg& reg = *w.reg;
    const gm_policy_t& pol = w.policy;
    for (auto& kv : reg.arm) {
      uint32_t gpu = kv.first;
      PolicyArm& a = kv.second;
      int64_t ue = 0, umc_ue = 0, gfx_ue = 0, replay = 0, xce = 0, xue = 0, down = 0;
      bool have_ue = get_i(gpu, GM_FI_ECC_UE_VOLATILE, &ue);
      get_i(gpu, GM_FI_ECC_UMC_UE, &umc_ue);
      get_i(gpu, GM_FI_ECC_GFX_UE, &gfx_ue);
      bool have_replay = get_i(gpu, GM_FI_PCIE_REPLAY_COUNTER, &replay);
      bool have_xce = get_i(gpu, GM_FI_XGMI_CRC_ERRORS, &xce);
      bool have_xue = get_i(gpu, GM_FI_XGMI_DATA_ERRORS, &xue);
      bool have_down = get_i(gpu, GM_FI_XGMI_LINKS_DOWN, &down);
      if (!a.baseline_valid) {
        if (!(have_ue || have_replay || have_xce || have_xue || have_down)) {
          // nothing sampled yet: level checks below still run
        } else {
          a.ue = ue; a.replay = replay; a.xgmi_ce = xce; a.xgmi_ue = xue; a.links_down = down;
          a.baseline_valid = true;
        }
      }
      gm_violation_t v;
      auto reset = [&](uint32_t cond) {
        memset(&v, 0, sizeof v);
        v.condition = cond;
        v.gpu = gpu;
        v.ts_us = now_wall;
      };

      if ((reg.conditions & GM_POLICY_COND_ECC_UE) && a.baseline_valid && have_ue && ue > a.ue) {
        reset(GM_POLICY_COND_ECC_UE);
        v.d.ecc.count = (uint32_t)(ue - a.ue);
        bool hbm = umc_ue > 0;
        v.d.ecc.block = hbm ? 0 : (gfx_ue > 0 ? 1 : 255);
        copy_str(v.d.ecc.location, std::string(hbm ? "HBM (UMC)" : (gfx_ue > 0 ? "GFX (XCD)" : "Device")));
        fire(w.reg, v);
      }
      if (have_ue && a.baseline_valid) a.ue = ue;

      if ((reg.conditions & GM_POLICY_COND_PCIE) && a.baseline_valid && have_replay && replay > a.replay) {
        reset(GM_POLICY_COND_PCIE);
        v.d.pcie.replay_counter = (uint32_t)replay;
        fire(w.reg, v);
      }
      if (have_replay && a.baseline_valid) a.replay = replay;

      if (reg.conditions & GM_POLICY_COND_MAX_BAD_PAGES) {
        int64_t ce = 0, uep = 0;
        bool have = get_i(gpu, GM_FI_BAD_PAGES_CE, &ce);
        have = get_i(gpu, GM_FI_BAD_PAGES_UE, &uep) || have;
        if (have) {
          bool over = ce + uep >= (int64_t)pol.max_bad_pages;
          if (over && !a.pages_fired) {
            reset(GM_POLICY_COND_MAX_BAD_PAGES);
            v.d.bad_pages.ce_pages = (uint32_t)ce;
            v.d.bad_pages.ue_pages = (uint32_t)uep;
            fire(w.reg, v);
          }
          a.pages_fired = over;
        }
      }

      if (reg.conditions & GM_POLICY_COND_THERMAL) {
        int64_t t;
        if (get_i(gpu, GM_FI_GPU_TEMP, &t)) {
          double rearm = (double)pol.thermal_c * (1.0 - pol.hysteresis_pct / 100.0);
          if (t >= (int64_t)pol.thermal_c && !a.thermal_fired) {
            reset(GM_POLICY_COND_THERMAL);
            v.d.thermal.temperature_c = (uint32_t)t;
            fire(w.reg, v);
            a.thermal_fired = true;
          } else if (a.thermal_fired && (double)t < rearm) {
            a.thermal_fired = false;
          }
        }
      }

      if (reg.conditions & GM_POLICY_COND_POWER) {
        double p;
        if (get_d(gpu, GM_FI_POWER_USAGE, &p)) {
          double rearm = (double)pol.power_w * (1.0 - pol.hysteresis_pct / 100.0);
          if (p >= (double)pol.power_w && !a.power_fired) {
            reset(GM_POLICY_COND_POWER);
            v.d.power.power_w = (uint32_t)p;
            fire(w.reg, v);
            a.power_fired = true;
          } else if (a.power_fired && p < rearm) {
            a.power_fired = false;
          }
        }
      }

      if ((reg.conditions & GM_POLICY_COND_XGMI) && a.baseline_valid) {
        if (have_xue && xue > a.xgmi_ue) {
          reset(GM_POLICY_COND_XGMI);
          v.d.xgmi.field_id = GM_FI_XGMI_DATA_ERRORS;
          v.d.xgmi.link = 0xffff;
          v.d.xgmi.counter = (uint32_t)xue;
          fire(w.reg, v);
        }
        if (have_xce && xce > a.xgmi_ce) {
          reset(GM_POLICY_COND_XGMI);
          v.d.xgmi.field_id = GM_FI_XGMI_CRC_ERRORS;
          v.d.xgmi.link = 0xffff;
          v.d.xgmi.counter = (uint32_t)xce;
          fire(w.reg, v);
        }
        if (have_down && down > a.links_down) {
          reset(GM_POLICY_COND_XGMI);
          v.d.xgmi.field_id = GM_FI_XGMI_LINKS_DOWN;
          v.d.xgmi.link = 0xffff;
          int64_t s;
          for (int l = 0; l < GM_MAX_XGMI_LINKS; l++)
            if (get_i(gpu, GM_FI_XGMI_LINK_STATUS_0 + l, &s) && s == 0) { v.d.xgmi.link = (uint16_t)l; break; }
          v.d.xgmi.counter = (uint32_t)down;
          fire(w.reg, v);
        }
      }
      if (a.baseline_valid) {
        if (have_xue) a.xgmi_ue = xue;
        if (have_xce) a.xgmi_ce = xce;
        if (have_down) a.links_down = down;
      }
    }
  }
}

}  // namespace gpumon
