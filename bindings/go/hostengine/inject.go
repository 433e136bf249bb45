package hostengine

/*
#include "capi.h"

static void he_set_value(gm_value_t* v, uint16_t field, uint8_t type, double x) {
  memset(v, 0, sizeof *v);
  v->field_id = field;
  v->type = type;
  v->status = GM_VS_OK;
  if (type == GM_FT_DOUBLE) v->v.dbl = x; else v->v.i64 = (int64_t)x;
}
*/
import "C"

// FakeCtl scripts the fake backend ("set gpu=0 field=150 value=95", "event gpu=1
// code=5", "link_down gpu=0 link=2" ...).  It returns a not-supported error on real
// hardware.  Demos and tests use it to raise health incidents and policy violations
// without touching a GPU.
func FakeCtl(command string) error {
	c, free := cstr(command)
	defer free()
	return check("fake ctl", C.gm_fake_ctl(c))
}

// InjectValue overrides the cached value of a numeric field for holdMs milliseconds
// (0: until the next sample) and re-evaluates policies at once; the `dcgmi test
// --inject` role, for demonstrating policies and health on healthy hardware.
func InjectValue(gpuID uint, field uint16, value float64, holdMs int) error {
	var m C.gm_field_meta_t
	if err := check("field meta", C.gm_field_meta(C.uint16_t(field), &m)); err != nil {
		return err
	}
	var v C.gm_value_t
	C.he_set_value(&v, C.uint16_t(field), m._type, C.double(value))
	return check("inject value", C.gm_inject_value(C.uint32_t(gpuID), &v, C.int32_t(holdMs)))
}
