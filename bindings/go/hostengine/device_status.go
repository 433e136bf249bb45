package hostengine

/*
#include "capi.h"
*/
import "C"

// DeviceStatus is the latest cached telemetry of one GPU.  The engine serves it
// from one persistent internal watch; nothing is created or destroyed per call
// (the reference builds and tears down a group, a field group and a watch on every
// GetDeviceStatus: bindings/go/dcgm/device_status.go:115-180).

// PerfState is the driver's performance level, the P-state analog (bindings/go/dcgm/device_status.go:14-27);
// the amdgpu driver has named levels instead of P0..P15.
type PerfState uint

const (
	PerfStateAuto PerfState = iota
	PerfStateLow
	PerfStateHigh
	PerfStateManual
	PerfStateStableStd
	PerfStateStablePeak
	PerfStateStableMinMclk
	PerfStateStableMinSclk
	PerfStateDeterminism
	PerfStateUnknown PerfState = 0x100
)

func (p PerfState) String() string {
	names := []string{"auto", "low", "high", "manual", "stable_std", "stable_peak", "stable_min_mclk", "stable_min_sclk", "determinism"}
	if int(p) < len(names) {
		return names[p]
	}
	return "Unknown"
}

type UtilizationInfo struct {
	GPU     *uint // %
	Memory  *uint // %
	Encoder *uint // nil on MI355X
	Decoder *uint
}

type ECCErrorsInfo struct {
	Correctable   *uint
	Uncorrectable *uint
}

type MemoryInfo struct {
	GlobalUsed *uint // MiB of HBM in use
	GlobalFree *uint
	ECCErrors  ECCErrorsInfo
}

type PCIThroughputInfo struct {
	Rx      *int64 // MB/s
	Tx      *int64
	Replays *int64
}

type PCIStatusInfo struct {
	VisibleVRAMUsed *uint // MiB
	Throughput      PCIThroughputInfo
}

type DeviceStatus struct {
	Power       *float64 // W
	Temperature *uint    // °C, hotspot
	HBMTemp     *uint    // °C
	Utilization UtilizationInfo
	Memory      MemoryInfo
	Clocks      ClockInfo
	PCI         PCIStatusInfo
	Performance *PerfState
	FanSpeed    *uint   // nil: OAM modules have no fan
	Throttle    *uint   // bit set of GM_THROTTLE_*
	EnergyJ     *float64
}

func latestValuesForDevice(gpuID uint) (DeviceStatus, error) {
	var s C.gm_device_status_t
	C.he_status(&s)
	if err := check("device status", C.gm_device_status(C.uint32_t(gpuID), &s)); err != nil {
		return DeviceStatus{}, err
	}
	ok := func(bit uint) bool { return uint64(s.valid)&(1<<bit) != 0 }
	kbToMB := func(v C.int64_t, valid bool) *int64 {
		if !valid {
			return nil
		}
		x := int64(v) / 1000
		return &x
	}
	st := DeviceStatus{
		Power:       optFloat(s.power_w, ok(C.GM_DS_POWER)),
		Temperature: optUint(s.temperature_c, ok(C.GM_DS_TEMP)),
		Utilization: UtilizationInfo{GPU: optUint(s.gfx_util, ok(C.GM_DS_GFX_UTIL)), Memory: optUint(s.mem_util, ok(C.GM_DS_MEM_UTIL)),
			Encoder: optUint(s.enc_util, ok(C.GM_DS_ENC_UTIL)), Decoder: optUint(s.dec_util, ok(C.GM_DS_DEC_UTIL))},
		Clocks: ClockInfo{Cores: optUint(s.gfx_clock_mhz, ok(6)), Memory: optUint(s.mem_clock_mhz, ok(7))},
		PCI: PCIStatusInfo{VisibleVRAMUsed: optUint(s.vis_vram_used_mib, ok(8)),
			Throughput: PCIThroughputInfo{Rx: kbToMB(s.pcie_rx_kb, ok(9)), Tx: kbToMB(s.pcie_tx_kb, ok(10)),
				Replays: optInt(s.pcie_replays, ok(11))}},
		Memory: MemoryInfo{GlobalUsed: optUint(s.vram_used_mib, ok(12)), GlobalFree: optUint(s.vram_free_mib, ok(13)),
			ECCErrors: ECCErrorsInfo{Correctable: optUint(s.ecc_ce, ok(14)), Uncorrectable: optUint(s.ecc_ue, ok(15))}},
		FanSpeed: optUint(s.fan_speed, ok(17)),
		Throttle: optUint(s.throttle_status, ok(18)),
		HBMTemp:  optUint(s.mem_temp_c, ok(19)),
	}
	if p := optUint(s.perf_level, ok(16)); p != nil {
		ps := PerfState(*p)
		st.Performance = &ps
	}
	if ok(20) {
		j := float64(s.energy_mj) / 1000
		st.EnergyJ = &j
	}
	return st, nil
}
