// Package hostengine is the Go client of the gpumon host engine: groups, field
// groups and watches, cached telemetry, health, policies with asynchronous
// violations, per-process accounting, topology and engine introspection.  The
// engine runs inside the process (Embedded), in a gpumon-hostengine daemon
// reached over a unix or TCP socket (Standalone), or in a private daemon this
// package spawns (StartHostengine).  It has the role and API shape of the
// reference's dcgm package (bindings/go/dcgm/*.go).
package hostengine

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../build -lgpumon -Wl,-rpath,${SRCDIR}/../../../build
#include "capi.h"
*/
import "C"

import (
	"fmt"
	"math"
	"unsafe"
)

// Error carries the library status of a failed call.
type Error struct {
	Status int
	Msg    string
	Op     string
}

func (e *Error) Error() string { return fmt.Sprintf("%s: %s (%d)", e.Op, e.Msg, e.Status) }

func check(op string, st C.gm_status_t) error {
	if st == C.GM_ST_OK {
		return nil
	}
	return &Error{Status: int(st), Msg: C.GoString(C.gm_strerror(st)), Op: op}
}

// IsStatus reports whether err is a library error with the given status code.
func IsStatus(err error, status int) bool {
	e, ok := err.(*Error)
	return ok && e.Status == status
}

const (
	StatusNotSupported    = int(C.GM_ST_NOT_SUPPORTED)
	StatusTimeout         = int(C.GM_ST_TIMEOUT)
	StatusNotFound        = int(C.GM_ST_NOT_FOUND)
	StatusNoData          = int(C.GM_ST_NO_DATA)
	StatusGpuNotSupported = int(C.GM_ST_GPU_NOT_SUPPORTED)
)

// Values are blank through an explicit status, never through sentinel numbers
// (the reference filters 0x7ffffff0-style blanks: bindings/go/dcgm/utils.go:99-125).

func optInt(v C.int64_t, valid bool) *int64 {
	if !valid {
		return nil
	}
	x := int64(v)
	return &x
}

func optUint(v C.int64_t, valid bool) *uint {
	if !valid || v < 0 {
		return nil
	}
	x := uint(v)
	return &x
}

func optFloat(v C.double, valid bool) *float64 {
	f := float64(v)
	if !valid || math.IsNaN(f) {
		return nil
	}
	return &f
}

func roundFloat(f *float64) *float64 {
	if f == nil {
		return nil
	}
	r := math.Round(*f)
	return &r
}

func cstr(s string) (*C.char, func()) {
	c := C.CString(s)
	return c, func() { C.free(unsafe.Pointer(c)) }
}

// Field ids used by this package (the full registry is in gpumon.h: GM_FI_*).
const (
	fiPCIeMaxGen   = uint16(C.GM_FI_DEV_PCIE_MAX_GEN)
	fiPCIeMaxWidth = uint16(C.GM_FI_DEV_PCIE_MAX_WIDTH)
)
