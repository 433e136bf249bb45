/* Helpers shared by the cgo files of this package: versioned-struct
 * initialisers and accessors for the unions cgo cannot address by member. */
#ifndef GPUMON_GO_CAPI_H_
#define GPUMON_GO_CAPI_H_

#include <stdlib.h>
#include <string.h>

#include "gpumon.h"

static inline void he_init_params(gm_init_params_t* p) { memset(p, 0, sizeof *p); p->version = GM_INIT_PARAMS_VERSION; }
static inline void he_attributes(gm_device_attributes_t* p) { memset(p, 0, sizeof *p); p->version = GM_DEVICE_ATTRIBUTES_VERSION; }
static inline void he_topology(gm_topology_t* p) { memset(p, 0, sizeof *p); p->version = GM_TOPOLOGY_VERSION; }
static inline void he_status(gm_device_status_t* p) { memset(p, 0, sizeof *p); p->version = GM_DEVICE_STATUS_VERSION; }
static inline void he_policy(gm_policy_t* p) { memset(p, 0, sizeof *p); p->version = GM_POLICY_VERSION; }
static inline void he_pid_info(gm_pid_info_t* p) { memset(p, 0, sizeof *p); p->version = GM_PID_INFO_VERSION; }
static inline void he_introspect(gm_introspect_t* p) { memset(p, 0, sizeof *p); p->version = GM_INTROSPECT_VERSION; }
static inline void he_health(gm_health_response_t* p, int n) {
  memset(p, 0, sizeof *p * (size_t)n);
  for (int i = 0; i < n; i++) p[i].version = GM_HEALTH_RESPONSE_VERSION;
}

static inline void he_thresholds(gm_health_thresholds_t* p) { memset(p, 0, sizeof *p); p->version = GM_HEALTH_THRESHOLDS_VERSION; }
static inline void he_diag_result(gm_diag_result_t* p) { memset(p, 0, sizeof *p); p->version = GM_DIAG_RESULT_VERSION; }
static inline void he_copy_str(char* dst, size_t cap, const char* src) {
  if (!cap) return;
  strncpy(dst, src, cap - 1);
  dst[cap - 1] = 0;
}

static inline int64_t he_value_i64(const gm_value_t* v) { return v->v.i64; }
static inline double he_value_dbl(const gm_value_t* v) { return v->v.dbl; }
static inline const char* he_value_str(const gm_value_t* v) { return v->v.str; }

/* gm_violation_t payload accessors */
static inline uint32_t he_v_ecc_block(const gm_violation_t* v) { return v->d.ecc.block; }
static inline uint32_t he_v_ecc_count(const gm_violation_t* v) { return v->d.ecc.count; }
static inline const char* he_v_ecc_location(const gm_violation_t* v) { return v->d.ecc.location; }
static inline uint32_t he_v_pcie_replays(const gm_violation_t* v) { return v->d.pcie.replay_counter; }
static inline uint32_t he_v_pages_ce(const gm_violation_t* v) { return v->d.bad_pages.ce_pages; }
static inline uint32_t he_v_pages_ue(const gm_violation_t* v) { return v->d.bad_pages.ue_pages; }
static inline uint32_t he_v_thermal(const gm_violation_t* v) { return v->d.thermal.temperature_c; }
static inline uint32_t he_v_power(const gm_violation_t* v) { return v->d.power.power_w; }
static inline uint32_t he_v_xgmi_field(const gm_violation_t* v) { return v->d.xgmi.field_id; }
static inline uint32_t he_v_xgmi_link(const gm_violation_t* v) { return v->d.xgmi.link; }
static inline uint32_t he_v_xgmi_counter(const gm_violation_t* v) { return v->d.xgmi.counter; }
static inline uint32_t he_v_event_code(const gm_violation_t* v) { return v->d.event.code; }
static inline const char* he_v_event_message(const gm_violation_t* v) { return v->d.event.message; }

/* callback.c */
int violationNotify(const gm_violation_t* v, void* user);
gm_status_t he_policy_register(gm_group_t g, uint32_t conditions, uintptr_t token, gm_policy_reg_t* out);

#endif
