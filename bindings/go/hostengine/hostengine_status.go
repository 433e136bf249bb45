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

// MetricsTableSource tells how the engine's device layer reads the bulk metrics table of a GPU: "direct" (its own
// read of the driver's gpu_metrics file, checked against the device library's decoding), "library", or
// "library: <why direct reads are off>"; "n/a" for backends without such a table.
func MetricsTableSource(gpu uint) (string, error) {
	var buf [256]C.char
	if err := check("metrics table source", C.gm_metrics_table_source(C.uint32_t(gpu), &buf[0], C.size_t(len(buf)))); err != nil {
		return "", err
	}
	return C.GoString(&buf[0]), nil
}
