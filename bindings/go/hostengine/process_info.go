package hostengine

/*
#include "capi.h"
*/
import "C"

import (
	"fmt"
	"time"
	"unsafe"
)

// Per-process accounting.  Field set of the reference's ProcessInfo
// (bindings/go/dcgm/process_info.go:18-62): run time, energy, average
// utilisation, PCIe averages, peak memory, ECC, average clocks, time spent under
// each throttle cause, and the time stamps of critical events ("XID errors").

type Time uint64

func (t Time) String() string {
	if t == 0 {
		return "Running"
	}
	return time.Unix(int64(t), 0).String()
}

type ProcessUtilInfo struct {
	StartTime      Time
	EndTime        Time
	EnergyConsumed *float64 // J
	GfxUtil        *float64 // the process' own share of the gfx engine, % ("SmUtil" in the reference)
	MemUtil        *float64 // the process' own share of the memory controllers, %
}

type PCIStats struct {
	AvgRxKBs *int64
	AvgTxKBs *int64
	Replays  *int64
}

type MemoryStats struct {
	GlobalUsed *int64 // peak bytes of HBM held
	ECCErrors  ECCErrorsInfo
}

type ViolationTime struct {
	Power          *time.Duration
	Thermal        *time.Duration
	Reliability    *time.Duration
	BoardLimit     *time.Duration
	LowUtilization *time.Duration
	SyncBoost      *time.Duration // always nil on AMD
}

type CriticalEvents struct {
	NumErrors int
	Timestamp []time.Time
}

type ProcessInfo struct {
	GPU                uint
	PID                uint
	Name               string
	ProcessUtilization ProcessUtilInfo
	PCI                PCIStats
	Memory             MemoryStats
	GpuUtilization     UtilizationInfo // of the whole device, averaged over the process lifetime (encoder / decoder stay nil)
	Clocks             ClockInfo       // averages over the process lifetime
	Violations         ViolationTime
	XIDErrors          CriticalEvents // resets, RAS errors and VM faults while the process ran: the XID analog
}

func watchPidFields() (GroupHandle, error) {
	g := GroupAllGPUs
	err := check("watch pid fields", C.gm_watch_pid_fields(g.handle, C.int64_t(DefaultUpdateFreq/time.Microsecond),
		C.double(DefaultMaxKeepAge.Seconds()), C.int32_t(DefaultMaxKeepSamples)))
	return g, err
}

func getProcessInfo(g GroupHandle, pid uint) ([]ProcessInfo, error) {
	info := (*C.gm_pid_info_t)(C.malloc(C.size_t(C.sizeof_gm_pid_info_t)))
	if info == nil {
		return nil, fmt.Errorf("out of memory")
	}
	defer C.free(unsafe.Pointer(info))
	C.he_pid_info(info)
	if err := check("pid info", C.gm_pid_info(g.handle, C.uint32_t(pid), info)); err != nil {
		return nil, err
	}
	out := make([]ProcessInfo, 0, int(info.num_gpus))
	for i := 0; i < int(info.num_gpus); i++ {
		s := &info.gpus[i]
		ok := func(bit uint) bool { return uint64(s.valid)&(1<<bit) != 0 }
		dur := func(v C.int64_t, bit uint) *time.Duration {
			if !ok(bit) {
				return nil
			}
			d := time.Duration(v) * time.Microsecond
			return &d
		}
		p := ProcessInfo{GPU: uint(s.gpu), PID: uint(s.pid), Name: C.GoString(&s.name[0])}
		p.ProcessUtilization = ProcessUtilInfo{StartTime: Time(s.start_ts_us / 1000000), EndTime: Time(s.end_ts_us / 1000000),
			GfxUtil: optFloat(s.proc_gfx_util_avg, ok(3)), MemUtil: optFloat(s.proc_mem_util_avg, ok(4))}
		p.GpuUtilization = UtilizationInfo{GPU: optUint(C.int64_t(float64(s.gfx_util_avg)+0.5), ok(1)),
			Memory: optUint(C.int64_t(float64(s.mem_util_avg)+0.5), ok(2))}
		if ok(0) {
			j := float64(s.energy_mj) / 1000
			p.ProcessUtilization.EnergyConsumed = &j
		}
		p.PCI = PCIStats{AvgRxKBs: optInt(s.pcie_rx_avg_kbs, ok(5)), AvgTxKBs: optInt(s.pcie_tx_avg_kbs, ok(6)),
			Replays: optInt(s.pcie_replays, ok(7))}
		p.Memory = MemoryStats{GlobalUsed: optInt(s.max_vram_bytes, ok(8)),
			ECCErrors: ECCErrorsInfo{Correctable: optUint(s.ecc_ce, ok(9)), Uncorrectable: optUint(s.ecc_ue, ok(10))}}
		p.Clocks = ClockInfo{Cores: optUint(s.gfx_clock_avg, ok(11)), Memory: optUint(s.mem_clock_avg, ok(12))}
		p.Violations = ViolationTime{Power: dur(s.power_violation_us, 13), Thermal: dur(s.thermal_violation_us, 14),
			Reliability: dur(s.reliability_violation_us, 15), BoardLimit: dur(s.board_limit_violation_us, 16),
			LowUtilization: dur(s.low_util_violation_us, 17), SyncBoost: dur(s.sync_boost_violation_us, 18)}
		p.XIDErrors.NumErrors = int(s.num_critical_events)
		for k := 0; k < int(s.num_critical_events) && k < int(C.GM_MAX_EVENT_TS); k++ {
			p.XIDErrors.Timestamp = append(p.XIDErrors.Timestamp, time.UnixMicro(int64(s.critical_event_ts_us[k])))
		}
		out = append(out, p)
	}
	return out, nil
}

// ProcessSource names where the process list of a GPU comes from on this host:
// "kfd" (the device library's table), "fdinfo" (per-client DRM statistics under
// /proc), "open-files" (holders of the render node: memory and engine time are
// then unknown), "fake" or "none".
func ProcessSource(gpu uint) (string, error) {
	var buf [32]C.char
	if err := check("process source", C.gm_process_source(C.uint32_t(gpu), &buf[0], C.size_t(len(buf)))); err != nil {
		return "", err
	}
	return C.GoString(&buf[0]), nil
}
