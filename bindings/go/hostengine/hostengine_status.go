package hostengine

/*
#include "capi.h"
*/
import "C"

import "time"

// Status is the engine's view of itself: the numbers behind the "<1 % of a core"
// budget.  Memory (KB) and CPU (%) are what the reference reports
// (bindings/go/dcgm/hostengine_status.go:18-49); the rest is sampler bookkeeping.
type Status struct {
	Memory        int64   // KB resident
	CPU           float64 // % of one core since introspection was enabled
	Uptime        time.Duration
	SampleCycles  uint64
	BackendErrors uint64
	LastCycle     time.Duration
	AvgCycle      time.Duration
}

func introspect() (Status, error) {
	if err := check("introspect toggle", C.gm_introspect_toggle(1)); err != nil {
		return Status{}, err
	}
	var in C.gm_introspect_t
	C.he_introspect(&in)
	if err := check("introspect", C.gm_introspect(&in)); err != nil {
		return Status{}, err
	}
	ms := func(v C.double) time.Duration { return time.Duration(float64(v) * float64(time.Millisecond)) }
	return Status{Memory: int64(in.memory_kb), CPU: float64(in.cpu_percent),
		Uptime: time.Duration(in.uptime_us) * time.Microsecond, SampleCycles: uint64(in.sample_cycles),
		BackendErrors: uint64(in.backend_errors), LastCycle: ms(in.last_cycle_ms), AvgCycle: ms(in.avg_cycle_ms)}, nil
}
