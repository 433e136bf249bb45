/* The C side of violation delivery.  Go functions cannot be passed as C function
 * pointers, so the library is given violationNotify, which forwards to the
 * exported Go function; `user` carries an integer token that selects the Go
 * registration (never a Go pointer).  Role of the reference's callback.c
 * (bindings/go/dcgm/callback.c:1-4). */
#include "capi.h"
#include "_cgo_export.h"

int violationNotify(const gm_violation_t* v, void* user) {
  return goViolationNotify((gm_violation_t*)v, (uintptr_t)user);
}

gm_status_t he_policy_register(gm_group_t g, uint32_t conditions, uintptr_t token, gm_policy_reg_t* out) {
  return gm_policy_register(g, conditions, violationNotify, (void*)token, out);
}
