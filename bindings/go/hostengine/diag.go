package hostengine

/*
#include "capi.h"
*/
import "C"

import (
	"math"
	"time"
	"unsafe"
)

// DiagResult is the verdict of one active diagnostic (gpumon-diag: memtest, HBM
// stream, MFMA / GEMM stress, link test) as the engine keeps it.  A failed result
// stays with the GPU until a later result of the same test replaces it; health
// checks report it under System and a failure raises a GPU event.  The reference
// has no counterpart: its health is passive only (bindings/go/dcgm/health.go:55-119).
type DiagResult struct {
	GPU     uint
	Test    string
	System  uint   // one Health* bit
	Health  string // "Healthy" / "Warning" / "Failure"
	Value   *float64
	Time    time.Time
	Message string
}

// Health levels for SubmitDiagResult.
const (
	DiagPass = int(C.GM_HEALTH_PASS)
	DiagWarn = int(C.GM_HEALTH_WARN)
	DiagFail = int(C.GM_HEALTH_FAIL)
)

// SubmitDiagResult hands a verdict to the engine; value may be NaN.
func SubmitDiagResult(gpuID uint, test string, system uint, health int, value float64, message string) error {
	var r C.gm_diag_result_t
	C.he_diag_result(&r)
	r.gpu = C.uint32_t(gpuID)
	r.system = C.uint32_t(system)
	r.health = C.int32_t(health)
	r.value = C.double(value)
	ct, cm := C.CString(test), C.CString(message)
	defer C.free(unsafe.Pointer(ct))
	defer C.free(unsafe.Pointer(cm))
	C.he_copy_str(&r.test[0], C.size_t(len(r.test)), ct)
	C.he_copy_str(&r.message[0], C.size_t(len(r.message)), cm)
	return check("diag result submit", C.gm_diag_result_submit(&r))
}

// GetDiagResults returns the latest verdict of every test that has reported for the GPU.
func GetDiagResults(gpuID uint) ([]DiagResult, error) {
	var raw [32]C.gm_diag_result_t
	n := C.int(len(raw))
	if err := check("diag results", C.gm_diag_results(C.uint32_t(gpuID), &raw[0], &n)); err != nil {
		return nil, err
	}
	out := make([]DiagResult, 0, int(n))
	for i := 0; i < int(n); i++ {
		d := &raw[i]
		var v *float64
		if f := float64(d.value); !math.IsNaN(f) {
			v = &f
		}
		out = append(out, DiagResult{GPU: uint(d.gpu), Test: C.GoString(&d.test[0]), System: uint(d.system),
			Health: C.GoString(C.gm_health_result_name(d.health)), Value: v, Time: time.UnixMicro(int64(d.ts_us)),
			Message: C.GoString(&d.message[0])})
	}
	return out, nil
}
