package hostengine

/*
#include "capi.h"
*/
import "C"

import (
	"fmt"
	"sync"
	"sync/atomic"
	"time"
)

// Policies: thresholds set on a group, violations delivered asynchronously.
// The seven conditions of the reference (bindings/go/dcgm/policy.go:23-31) map to
// the AMD set: DbePolicy -> uncorrectable RAS error, PCIePolicy -> PCIe replay,
// MaxRtPgPolicy -> bad-page count, ThermalPolicy, PowerPolicy, NvlinkPolicy ->
// xGMI error / link down, XidPolicy -> critical GPU event (reset, VM fault, RAS).

type policyCondition string

const (
	EccUePolicy    = policyCondition("Uncorrectable ECC error")
	PCIePolicy     = policyCondition("PCI error")
	MaxBadPgPolicy = policyCondition("Max Bad Pages Limit")
	ThermalPolicy  = policyCondition("Thermal Limit")
	PowerPolicy    = policyCondition("Power Limit")
	XgmiPolicy     = policyCondition("xGMI error")
	GpuEventPolicy = policyCondition("GPU event")
)

var conditionBits = map[policyCondition]uint32{
	EccUePolicy: C.GM_POLICY_COND_ECC_UE, PCIePolicy: C.GM_POLICY_COND_PCIE, MaxBadPgPolicy: C.GM_POLICY_COND_MAX_BAD_PAGES,
	ThermalPolicy: C.GM_POLICY_COND_THERMAL, PowerPolicy: C.GM_POLICY_COND_POWER, XgmiPolicy: C.GM_POLICY_COND_XGMI,
	GpuEventPolicy: C.GM_POLICY_COND_GPU_EVENT,
}

// Thresholds of the level-triggered conditions.  The reference hard-codes 10 pages,
// 100 °C and 250 W (policy.go:118-120); here they are parameters with defaults sized
// for an MI355X (1400 W board power).
type Thresholds struct {
	MaxBadPages   uint
	ThermalC      uint
	PowerW        uint
	HysteresisPct uint
}

func DefaultThresholds() Thresholds {
	return Thresholds{MaxBadPages: 10, ThermalC: 100, PowerW: 1400, HysteresisPct: 3}
}

// Payloads, one per condition.
type EccUeData struct {
	Block    uint
	Count    uint
	Location string
}
type PCIeData struct{ ReplayCounter uint }
type BadPagesData struct{ CorrectablePages, UncorrectablePages uint }
type ThermalData struct{ ThermalViolation uint }
type PowerData struct{ PowerViolation uint }
type XgmiData struct {
	FieldID uint16
	Link    uint
	Counter uint
}
type GpuEventData struct {
	Code    uint
	Name    string
	Message string
}

type PolicyViolation struct {
	GPU       uint
	Condition policyCondition
	Timestamp time.Time
	Data      interface{}
}

// One listener per registration: token -> fan-out of its subscribers.
type listener struct {
	reg   C.gm_policy_reg_t
	group GroupHandle
	out   *fanout
}

var (
	listenersMu sync.Mutex
	listeners   = map[uintptr]*listener{}
	nextToken   uintptr
	policySeq   uint64
)

//export goViolationNotify
func goViolationNotify(v *C.gm_violation_t, token C.uintptr_t) C.int {
	listenersMu.Lock()
	l := listeners[uintptr(token)]
	listenersMu.Unlock()
	if l == nil {
		return 0
	}
	pv := PolicyViolation{GPU: uint(v.gpu), Timestamp: time.UnixMicro(int64(v.ts_us))}
	switch uint32(v.condition) {
	case C.GM_POLICY_COND_ECC_UE:
		pv.Condition = EccUePolicy
		pv.Data = EccUeData{Block: uint(C.he_v_ecc_block(v)), Count: uint(C.he_v_ecc_count(v)),
			Location: C.GoString(C.he_v_ecc_location(v))}
	case C.GM_POLICY_COND_PCIE:
		pv.Condition = PCIePolicy
		pv.Data = PCIeData{ReplayCounter: uint(C.he_v_pcie_replays(v))}
	case C.GM_POLICY_COND_MAX_BAD_PAGES:
		pv.Condition = MaxBadPgPolicy
		pv.Data = BadPagesData{CorrectablePages: uint(C.he_v_pages_ce(v)), UncorrectablePages: uint(C.he_v_pages_ue(v))}
	case C.GM_POLICY_COND_THERMAL:
		pv.Condition = ThermalPolicy
		pv.Data = ThermalData{ThermalViolation: uint(C.he_v_thermal(v))}
	case C.GM_POLICY_COND_POWER:
		pv.Condition = PowerPolicy
		pv.Data = PowerData{PowerViolation: uint(C.he_v_power(v))}
	case C.GM_POLICY_COND_XGMI:
		pv.Condition = XgmiPolicy
		pv.Data = XgmiData{FieldID: uint16(C.he_v_xgmi_field(v)), Link: uint(C.he_v_xgmi_link(v)), Counter: uint(C.he_v_xgmi_counter(v))}
	case C.GM_POLICY_COND_GPU_EVENT:
		code := uint(C.he_v_event_code(v))
		pv.Condition = GpuEventPolicy
		pv.Data = GpuEventData{Code: code, Name: C.GoString(C.gm_event_code_name(C.uint32_t(code))),
			Message: C.GoString(C.he_v_event_message(v))}
	default:
		return 0
	}
	l.out.publish(pv)
	return 0
}

func registerPolicy(gpuID uint, t Thresholds, conditions ...policyCondition) (<-chan PolicyViolation, error) {
	var mask uint32
	if len(conditions) == 0 {
		mask = C.GM_POLICY_COND_ALL
	}
	for _, c := range conditions {
		bit, ok := conditionBits[c]
		if !ok {
			return nil, fmt.Errorf("unknown policy condition %q", c)
		}
		mask |= bit
	}
	g, err := CreateGroup(fmt.Sprintf("go-policy-%d-%d", gpuID, atomic.AddUint64(&policySeq, 1)))
	if err != nil {
		return nil, err
	}
	fail := func(err error) (<-chan PolicyViolation, error) {
		_ = DestroyGroup(g)
		return nil, err
	}
	if err = AddToGroup(g, gpuID); err != nil {
		return fail(err)
	}
	var p C.gm_policy_t
	C.he_policy(&p)
	p.conditions = C.uint32_t(mask)
	p.max_bad_pages, p.thermal_c = C.uint32_t(t.MaxBadPages), C.uint32_t(t.ThermalC)
	p.power_w, p.hysteresis_pct = C.uint32_t(t.PowerW), C.uint32_t(t.HysteresisPct)
	if err = check("policy set", C.gm_policy_set(g.handle, &p)); err != nil {
		return fail(err)
	}
	l := &listener{group: g, out: newFanout()}
	ch := l.out.subscribe()
	listenersMu.Lock()
	nextToken++
	token := nextToken
	listeners[token] = l
	listenersMu.Unlock()
	if err = check("policy register", C.he_policy_register(g.handle, C.uint32_t(mask), C.uintptr_t(token), &l.reg)); err != nil {
		listenersMu.Lock()
		delete(listeners, token)
		listenersMu.Unlock()
		return fail(err)
	}
	return ch, nil
}

// stopAllPolicyListeners unregisters everything; called when the last Init is released.
func stopAllPolicyListeners() {
	listenersMu.Lock()
	all := listeners
	listeners = map[uintptr]*listener{}
	listenersMu.Unlock()
	for _, l := range all {
		// After unregister returns the library runs no further callback of this registration.
		_ = check("policy unregister", C.gm_policy_unregister(l.reg))
		_ = DestroyGroup(l.group)
		l.out.closeAll()
	}
}
