package hostengine

/*
#include "capi.h"
*/
import "C"

import "unsafe"

// Run modes of Init (the reference's Embedded / Standalone / StartHostengine:
// bindings/go/dcgm/admin.go:26-30).
type mode int

const (
	// Embedded runs the engine (sampler, health, policies) inside this process.
	Embedded mode = iota
	// Standalone connects to a running gpumon-hostengine; args: address, "1" when
	// the address is a unix socket path ("0" or absent: host:port).
	Standalone
	// StartHostengine spawns a private gpumon-hostengine on a temporary unix socket,
	// in its own process group, and stops it on Shutdown; args: optional path of the binary.
	StartHostengine
)

func initEngine(m mode, args ...string) error {
	var p C.gm_init_params_t
	C.he_init_params(&p)
	var frees []func()
	defer func() {
		for _, f := range frees {
			f()
		}
	}()
	hold := func(s string) *C.char {
		c := C.CString(s)
		frees = append(frees, func() { C.free(unsafe.Pointer(c)) })
		return c
	}
	switch m {
	case Embedded:
		p.mode = C.GM_MODE_EMBEDDED
		if len(args) > 0 && args[0] != "" { // backend override, e.g. "fake"
			p.backend = hold(args[0])
		}
		if len(args) > 1 {
			p.backend_spec = hold(args[1])
		}
	case Standalone:
		if len(args) < 1 {
			return &Error{Status: int(C.GM_ST_BADPARAM), Msg: "missing host engine address", Op: "init"}
		}
		p.mode = C.GM_MODE_STANDALONE
		p.address = hold(args[0])
		if len(args) > 1 && args[1] == "1" {
			p.address_is_unix_socket = 1
		}
	case StartHostengine:
		p.mode = C.GM_MODE_START_HOSTENGINE
		if len(args) > 0 && args[0] != "" {
			p.hostengine_path = hold(args[0])
		}
		if len(args) > 1 && args[1] != "" {
			p.backend = hold(args[1])
		}
		if len(args) > 2 {
			p.backend_spec = hold(args[2])
		}
	}
	return check("init", C.gm_init(&p))
}

func shutdownEngine() error { return check("shutdown", C.gm_shutdown()) }
