package hostengine

/*
#include "capi.h"
*/
import "C"

import "fmt"

// Health results: "Healthy" / "Warning" / "Failure" exactly as the reference prints
// them (bindings/go/dcgm/health.go:88-98); subsystems are the AMD set — PCIe, xGMI,
// Memory (HBM ECC and bad pages), Compute (XCD RAS), Thermal, Power, Driver, Firmware.

type SystemWatch struct {
	Type   string
	Status string
	Error  string
}

type DeviceHealth struct {
	GPU     uint
	Status  string
	Watches []SystemWatch
}

// Health watch subsystems for SetHealthWatches.
const (
	HealthPCIe     = uint(C.GM_HEALTH_WATCH_PCIE)
	HealthXGMI     = uint(C.GM_HEALTH_WATCH_XGMI)
	HealthMemory   = uint(C.GM_HEALTH_WATCH_MEM)
	HealthCompute  = uint(C.GM_HEALTH_WATCH_COMPUTE)
	HealthThermal  = uint(C.GM_HEALTH_WATCH_THERMAL)
	HealthPower    = uint(C.GM_HEALTH_WATCH_POWER)
	HealthDriver   = uint(C.GM_HEALTH_WATCH_DRIVER)
	HealthFirmware = uint(C.GM_HEALTH_WATCH_FIRMWARE)
	HealthAll      = uint(C.GM_HEALTH_WATCH_ALL)
)

func SetHealthWatches(g GroupHandle, systems uint) error {
	return check("health set", C.gm_health_set(g.handle, C.uint32_t(systems)))
}

// HealthCheck runs the enabled watches on every GPU of the group.
func HealthCheck(g GroupHandle) ([]DeviceHealth, error) {
	buf := make([]C.gm_health_response_t, C.GM_MAX_GPUS)
	C.he_health(&buf[0], C.int(len(buf)))
	n := C.int(len(buf))
	if err := check("health check", C.gm_health_check(g.handle, &buf[0], &n)); err != nil {
		return nil, err
	}
	out := make([]DeviceHealth, 0, int(n))
	for i := 0; i < int(n); i++ {
		r := &buf[i]
		h := DeviceHealth{GPU: uint(r.gpu), Status: C.GoString(C.gm_health_result_name(r.overall))}
		for k := 0; k < int(r.incident_count); k++ {
			inc := &r.incidents[k]
			h.Watches = append(h.Watches, SystemWatch{Type: C.GoString(C.gm_health_system_name(inc.system)),
				Status: C.GoString(C.gm_health_result_name(inc.health)), Error: C.GoString(&inc.message[0])})
		}
		out = append(out, h)
	}
	return out, nil
}

// One group per GPU, created on first use and kept: counters are judged against the
// baseline taken when the watch was set, so the group must outlive a single check.
var healthGroups = map[uint]GroupHandle{}

func healthCheckByGpuID(gpuID uint) (DeviceHealth, error) {
	initMu.Lock()
	g, ok := healthGroups[gpuID]
	if !ok {
		var err error
		if g, err = CreateGroup(fmt.Sprintf("go-health-%d", gpuID)); err != nil {
			initMu.Unlock()
			return DeviceHealth{}, err
		}
		if err = AddToGroup(g, gpuID); err == nil {
			err = SetHealthWatches(g, HealthAll)
		}
		if err != nil {
			_ = DestroyGroup(g)
			initMu.Unlock()
			return DeviceHealth{}, err
		}
		healthGroups[gpuID] = g
	}
	initMu.Unlock()
	res, err := HealthCheck(g)
	if err != nil {
		return DeviceHealth{}, err
	}
	if len(res) != 1 {
		return DeviceHealth{}, fmt.Errorf("health check returned %d results for one GPU", len(res))
	}
	return res[0], nil
}

// HealthThresholds are the levels the health checks compare against.  The reference has
// no such knob: its levels live inside the closed engine.  Defaults suit MI355X-class parts.
type HealthThresholds struct {
	HotspotWarnC, HotspotFailC int
	MemWarnC, MemFailC         int
	BadPagesFail               int
	CeBurstWarn                int
	PowerOverCapWarn           float64
}

func GetHealthThresholds() (HealthThresholds, error) {
	var t C.gm_health_thresholds_t
	C.he_thresholds(&t)
	if err := check("health thresholds get", C.gm_health_thresholds_get(&t)); err != nil {
		return HealthThresholds{}, err
	}
	return HealthThresholds{HotspotWarnC: int(t.hotspot_warn_c), HotspotFailC: int(t.hotspot_fail_c), MemWarnC: int(t.mem_warn_c),
		MemFailC: int(t.mem_fail_c), BadPagesFail: int(t.bad_pages_fail), CeBurstWarn: int(t.ce_burst_warn),
		PowerOverCapWarn: float64(t.power_over_cap_warn)}, nil
}

func SetHealthThresholds(h HealthThresholds) error {
	var t C.gm_health_thresholds_t
	C.he_thresholds(&t)
	t.hotspot_warn_c, t.hotspot_fail_c = C.int32_t(h.HotspotWarnC), C.int32_t(h.HotspotFailC)
	t.mem_warn_c, t.mem_fail_c = C.int32_t(h.MemWarnC), C.int32_t(h.MemFailC)
	t.bad_pages_fail, t.ce_burst_warn = C.int32_t(h.BadPagesFail), C.int32_t(h.CeBurstWarn)
	t.power_over_cap_warn = C.double(h.PowerOverCapWarn)
	return check("health thresholds set", C.gm_health_thresholds_set(&t))
}
