package hostengine

/*
#include "capi.h"
*/
import "C"

import "time"

// Watch defaults: 1 s period, 300 s retention, no sample cap — the operating point
// of the reference (bindings/go/dcgm/fields.go:12-16), now parameters instead of constants.
const (
	DefaultUpdateFreq     = time.Second
	DefaultMaxKeepAge     = 300 * time.Second
	DefaultMaxKeepSamples = 0
)

// FieldHandle names a set of field ids inside the engine.
type FieldHandle struct{ handle C.gm_field_group_t }

// Value is one cached sample; V is int64, float64, string, or nil when blank.
type Value struct {
	FieldID uint16
	Time    time.Time
	V       interface{}
}

func FieldGroupCreate(name string, fields []uint16) (FieldHandle, error) {
	c, free := cstr(name)
	defer free()
	ids := make([]C.uint16_t, len(fields))
	for i, f := range fields {
		ids[i] = C.uint16_t(f)
	}
	var fg C.gm_field_group_t
	var p *C.uint16_t
	if len(ids) > 0 {
		p = &ids[0]
	}
	err := check("field group create", C.gm_field_group_create(p, C.int(len(ids)), c, &fg))
	return FieldHandle{fg}, err
}

func FieldGroupDestroy(f FieldHandle) error {
	return check("field group destroy", C.gm_field_group_destroy(f.handle))
}

// WatchFields starts sampling the fields on the group's GPUs with the default period and retention.
func WatchFields(g GroupHandle, f FieldHandle) error {
	return WatchFieldsWithOptions(g, f, DefaultUpdateFreq, DefaultMaxKeepAge, DefaultMaxKeepSamples)
}

func WatchFieldsWithOptions(g GroupHandle, f FieldHandle, freq, keepAge time.Duration, keepSamples int) error {
	return check("watch fields", C.gm_watch_fields(g.handle, f.handle, C.int64_t(freq/time.Microsecond),
		C.double(keepAge.Seconds()), C.int32_t(keepSamples)))
}

func UnwatchFields(g GroupHandle, f FieldHandle) error {
	return check("unwatch fields", C.gm_unwatch_fields(g.handle, f.handle))
}

// UpdateAllFields forces one sample cycle of everything watched and waits for it.
func UpdateAllFields() error { return check("update all fields", C.gm_update_all_fields(1)) }

func decodeValue(v *C.gm_value_t) Value {
	out := Value{FieldID: uint16(v.field_id), Time: time.UnixMicro(int64(v.ts_us))}
	if v.status != C.GM_VS_OK {
		return out
	}
	switch v._type {
	case C.GM_FT_INT64:
		out.V = int64(C.he_value_i64(v))
	case C.GM_FT_DOUBLE:
		out.V = float64(C.he_value_dbl(v))
	case C.GM_FT_STRING:
		out.V = C.GoString(C.he_value_str(v))
	}
	return out
}

// GetLatestValues reads the newest cached sample of each field; one consistent snapshot per call.
func GetLatestValues(gpuID uint, fields []uint16) ([]Value, error) {
	if len(fields) == 0 {
		return nil, nil
	}
	ids := make([]C.uint16_t, len(fields))
	for i, f := range fields {
		ids[i] = C.uint16_t(f)
	}
	vals := make([]C.gm_value_t, len(fields))
	if err := check("latest values", C.gm_latest_values(C.uint32_t(gpuID), &ids[0], C.int(len(ids)), &vals[0])); err != nil {
		return nil, err
	}
	out := make([]Value, len(vals))
	for i := range vals {
		out[i] = decodeValue(&vals[i])
	}
	return out, nil
}

// GetValuesSince returns the retained samples of one field newer than since.
func GetValuesSince(gpuID uint, field uint16, since time.Time) ([]Value, error) {
	buf := make([]C.gm_value_t, 4096)
	n := C.int(len(buf))
	ts := C.int64_t(0)
	if !since.IsZero() {
		ts = C.int64_t(since.UnixMicro())
	}
	st := C.gm_values_since(C.uint32_t(gpuID), C.uint16_t(field), ts, &buf[0], &n)
	if st != C.GM_ST_OK && st != C.GM_ST_INSUFFICIENT_SIZE {
		return nil, check("values since", st)
	}
	if int(n) > len(buf) {
		n = C.int(len(buf))
	}
	out := make([]Value, int(n))
	for i := range out {
		out[i] = decodeValue(&buf[i])
	}
	return out, nil
}
