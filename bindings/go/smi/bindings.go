// Package smi is the direct device layer for AMD Instinct GPUs: every call is one
// synchronous round to the device library (libamd_smi.so, loaded lazily by
// libgpumon), with no host engine and no cache.  It has the role and the API
// shape of the reference's nvml package (bindings/go/nvml/bindings.go and
// nvml.go), with AMD semantics: Path is the DRM render node, links are xGMI,
// optional values are nil when the device does not report them.
//
// This file holds the thin cgo calls; smi.go builds the public types on top.
package smi

/*
#cgo CFLAGS: -I${SRCDIR}/../../../include
#cgo LDFLAGS: -L${SRCDIR}/../../../build -lgpumon -Wl,-rpath,${SRCDIR}/../../../build
#include <stdlib.h>
#include <string.h>
#include "gpumon.h"

static void smi_attr_init(gm_device_attributes_t* a) { memset(a, 0, sizeof *a); a->version = GM_DEVICE_ATTRIBUTES_VERSION; }
static void smi_status_init(gm_device_status_t* s) { memset(s, 0, sizeof *s); s->version = GM_DEVICE_STATUS_VERSION; }
static void smi_mode_init(gm_device_mode_t* m) { memset(m, 0, sizeof *m); m->version = GM_DEVICE_MODE_VERSION; }
static int64_t smi_value_i64(const gm_value_t* v) { return v->v.i64; }
static double smi_value_dbl(const gm_value_t* v) { return v->v.dbl; }
static const char* smi_value_str(const gm_value_t* v) { return v->v.str; }
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Error carries the library status of a failed call.
type Error struct {
	Status int
	Msg    string
	Op     string
}

func (e *Error) Error() string { return fmt.Sprintf("%s: %s (%d)", e.Op, e.Msg, e.Status) }

// Library status codes callers commonly branch on.
const (
	StatusOK            = int(C.GM_ST_OK)
	StatusNotSupported  = int(C.GM_ST_NOT_SUPPORTED)
	StatusNoPermission  = int(C.GM_ST_NO_PERMISSION)
	StatusUninitialized = int(C.GM_ST_UNINITIALIZED)
	StatusTimeout       = int(C.GM_ST_TIMEOUT)
	StatusGpuUnsupport  = int(C.GM_ST_GPU_NOT_SUPPORTED)
)

func check(op string, st C.gm_status_t) error {
	if st == C.GM_ST_OK {
		return nil
	}
	return &Error{Status: int(st), Msg: C.GoString(C.gm_strerror(st)), Op: op}
}

func isStatus(err error, status int) bool {
	e, ok := err.(*Error)
	return ok && e.Status == status
}

func cstrOrNil(s string) *C.char {
	if s == "" {
		return nil
	}
	return C.CString(s)
}

func initLib(backend, spec string) error {
	b, s := cstrOrNil(backend), cstrOrNil(spec)
	defer C.free(unsafe.Pointer(b))
	defer C.free(unsafe.Pointer(s))
	return check("smi init", C.gm_smi_init(b, s))
}

func shutdownLib() error { return check("smi shutdown", C.gm_smi_shutdown()) }

func deviceCount() (uint, error) {
	var n C.uint32_t
	err := check("device count", C.gm_smi_device_count(&n))
	return uint(n), err
}

func driverVersion() (string, error) {
	var buf [C.GM_STR_LEN]C.char
	if err := check("driver version", C.gm_smi_driver_version(&buf[0], C.size_t(len(buf)))); err != nil {
		return "", err
	}
	return C.GoString(&buf[0]), nil
}

// runtimeVersion: ROCm release under the device library; nil, nil when it cannot be told
// (the reference maps NVML_ERROR_NOT_SUPPORTED the same way, bindings/go/nvml/bindings.go:152-165).
func runtimeVersion() (*uint, *uint, string, error) {
	var major, minor C.uint32_t
	var buf [C.GM_STR_LEN]C.char
	st := C.gm_smi_runtime_version(&major, &minor, &buf[0], C.size_t(len(buf)))
	if st == C.GM_ST_NOT_SUPPORTED {
		return nil, nil, "", nil
	}
	if err := check("runtime version", st); err != nil {
		return nil, nil, "", err
	}
	ma, mi := uint(major), uint(minor)
	return &ma, &mi, C.GoString(&buf[0]), nil
}

type rawAttributes struct {
	supported                              bool
	uuid, name, brand, serial, vbios       string
	firmware, driver, busID, gfxTarget     string
	path, cpuAffinity, computePart, memPart string
	numaNode, hipID, renderMinor           int
	numXCD, numCU, pcieGen, pcieWidth      uint
	pcieBandwidthMBs                       int64
	powerCapW, powerCapDefaultW            float64
	vramTotalMiB, visVramTotalMiB          int64
	gfxClockMaxMHz, memClockMaxMHz         int64
}

func deviceAttributes(idx uint) (*rawAttributes, error) {
	var a C.gm_device_attributes_t
	C.smi_attr_init(&a)
	if err := check("device attributes", C.gm_smi_device_attributes(C.uint32_t(idx), &a)); err != nil {
		return nil, err
	}
	return &rawAttributes{
		supported: a.supported != 0,
		uuid:      C.GoString(&a.uuid[0]), name: C.GoString(&a.name[0]), brand: C.GoString(&a.brand[0]),
		serial: C.GoString(&a.serial[0]), vbios: C.GoString(&a.vbios[0]), firmware: C.GoString(&a.firmware[0]),
		driver: C.GoString(&a.driver_version[0]), busID: C.GoString(&a.pci_bdf[0]),
		gfxTarget: C.GoString(&a.gfx_target[0]), path: C.GoString(&a.device_path[0]),
		cpuAffinity: C.GoString(&a.cpu_affinity[0]), computePart: C.GoString(&a.compute_partition[0]),
		memPart: C.GoString(&a.memory_partition[0]), numaNode: int(a.numa_node), hipID: int(a.hip_id),
		renderMinor: int(a.drm_render_minor), numXCD: uint(a.num_xcd), numCU: uint(a.num_cu),
		pcieGen: uint(a.pcie_max_gen), pcieWidth: uint(a.pcie_max_width),
		pcieBandwidthMBs: int64(a.pcie_bandwidth_mbs), powerCapW: float64(a.power_cap_w),
		powerCapDefaultW: float64(a.power_cap_default_w), vramTotalMiB: int64(a.vram_total_mib),
		visVramTotalMiB: int64(a.vis_vram_total_mib), gfxClockMaxMHz: int64(a.gfx_clock_max_mhz),
		memClockMaxMHz: int64(a.mem_clock_max_mhz),
	}, nil
}

// rawStatus mirrors gm_device_status_t; valid holds one bit per member in declaration order.
type rawStatus struct {
	valid                                                              uint64
	powerW                                                             float64
	temp, gfxUtil, memUtil, encUtil, decUtil, gfxClock, memClock       int64
	visVramUsed, pcieRxKB, pcieTxKB, pcieReplays, vramUsed, vramFree   int64
	eccCE, eccUE, perfLevel, fanSpeed, throttle, memTemp, energyMilliJ int64
}

func deviceStatus(idx uint) (*rawStatus, error) {
	var s C.gm_device_status_t
	C.smi_status_init(&s)
	if err := check("device status", C.gm_smi_device_status(C.uint32_t(idx), &s)); err != nil {
		return nil, err
	}
	return &rawStatus{
		valid: uint64(s.valid), powerW: float64(s.power_w), temp: int64(s.temperature_c),
		gfxUtil: int64(s.gfx_util), memUtil: int64(s.mem_util), encUtil: int64(s.enc_util), decUtil: int64(s.dec_util),
		gfxClock: int64(s.gfx_clock_mhz), memClock: int64(s.mem_clock_mhz), visVramUsed: int64(s.vis_vram_used_mib),
		pcieRxKB: int64(s.pcie_rx_kb), pcieTxKB: int64(s.pcie_tx_kb), pcieReplays: int64(s.pcie_replays),
		vramUsed: int64(s.vram_used_mib), vramFree: int64(s.vram_free_mib), eccCE: int64(s.ecc_ce), eccUE: int64(s.ecc_ue),
		perfLevel: int64(s.perf_level), fanSpeed: int64(s.fan_speed), throttle: int64(s.throttle_status),
		memTemp: int64(s.mem_temp_c), energyMilliJ: int64(s.energy_mj),
	}, nil
}

// GM_PROC_MEM_UNKNOWN of gpumon.h.
const procMemUnknown = ^uint64(0)

type rawProcess struct {
	pid         uint
	name        string
	vramBytes   uint64
	gfxEngineNs uint64
	cuOccupancy uint
}

func deviceProcesses(idx uint) ([]rawProcess, error) {
	buf := make([]C.gm_process_t, C.GM_MAX_PROCS)
	n := C.int(len(buf))
	st := C.gm_smi_device_processes(C.uint32_t(idx), &buf[0], &n)
	if st == C.GM_ST_NOT_SUPPORTED || st == C.GM_ST_NO_PERMISSION {
		return nil, nil // no process table for this user: "none", like the other optional values
	}
	if st != C.GM_ST_OK && st != C.GM_ST_INSUFFICIENT_SIZE {
		return nil, check("device processes", st)
	}
	if int(n) > len(buf) {
		n = C.int(len(buf))
	}
	out := make([]rawProcess, 0, int(n))
	for i := 0; i < int(n); i++ {
		p := &buf[i]
		out = append(out, rawProcess{pid: uint(p.pid), name: C.GoString(&p.name[0]), vramBytes: uint64(p.vram_bytes),
			gfxEngineNs: uint64(p.gfx_engine_ns), cuOccupancy: uint(p.cu_occupancy)})
	}
	return out, nil
}

type rawPeerLink struct {
	link, hops, numLinks uint
	p2p                  bool
	busID                string
	maxBandwidthMBs      int64
}

func peerLink(a, b uint) (*rawPeerLink, error) {
	var p C.gm_peer_link_t
	if err := check("peer link", C.gm_smi_peer_link(C.uint32_t(a), C.uint32_t(b), &p)); err != nil {
		return nil, err
	}
	return &rawPeerLink{link: uint(p.link), hops: uint(p.hops), numLinks: uint(p.num_links), p2p: p.p2p_accessible != 0,
		busID: C.GoString(&p.pci_bdf[0]), maxBandwidthMBs: int64(p.max_bw_mbs)}, nil
}

type rawXgmiLink struct {
	link                 uint
	state, peer          int
	peerBusID            string
	bitRate, maxBw       int64
	rxKB, txKB           int64
}

func xgmiLinks(idx uint) ([]rawXgmiLink, error) {
	var raw [C.GM_MAX_XGMI_LINKS]C.gm_xgmi_link_t
	n := C.int(C.GM_MAX_XGMI_LINKS)
	if err := check("xgmi links", C.gm_smi_xgmi_links(C.uint32_t(idx), &raw[0], &n)); err != nil {
		return nil, err
	}
	out := make([]rawXgmiLink, 0, int(n))
	for i := 0; i < int(n); i++ {
		l := &raw[i]
		out = append(out, rawXgmiLink{link: uint(l.link), state: int(l.state), peer: int(l.peer_gpu), peerBusID: C.GoString(&l.peer_bdf[0]),
			bitRate: int64(l.bit_rate_gbps), maxBw: int64(l.max_bandwidth_gbps), rxKB: int64(l.rx_kb), txKB: int64(l.tx_kb)})
	}
	return out, nil
}

func linkName(link, hops uint) string { return C.GoString(C.gm_link_type_name(C.int32_t(link), C.uint32_t(hops))) }
func linkDescription(link uint) string { return C.GoString(C.gm_link_type_description(C.int32_t(link))) }

type rawMode struct {
	compute, memory      string
	accountingOn         bool
	accountingBufferSize uint
}

func deviceMode(idx uint) (rawMode, error) {
	var m C.gm_device_mode_t
	C.smi_mode_init(&m)
	if err := check("device mode", C.gm_smi_device_mode(C.uint32_t(idx), &m)); err != nil {
		return rawMode{}, err
	}
	return rawMode{compute: C.GoString(&m.compute_partition[0]), memory: C.GoString(&m.memory_partition[0]),
		accountingOn: m.accounting_enabled != 0, accountingBufferSize: uint(m.accounting_buffer_size)}, nil
}

type rawEvent struct {
	gpu, code uint
	tsMicros  int64
	uuid, msg string
}

func eventWait(timeoutMs int) (*rawEvent, error) {
	var e C.gm_event_t
	if err := check("event wait", C.gm_smi_event_wait(C.int32_t(timeoutMs), &e)); err != nil {
		return nil, err
	}
	return &rawEvent{gpu: uint(e.gpu), code: uint(e.code), tsMicros: int64(e.ts_us), uuid: C.GoString(&e.uuid[0]),
		msg: C.GoString(&e.message[0])}, nil
}

func eventName(code uint) string { return C.GoString(C.gm_event_code_name(C.uint32_t(code))) }

// readFields returns one optional value per requested field id: int64, float64, string or nil.
func readFields(idx uint, ids []uint16) ([]interface{}, error) {
	if len(ids) == 0 {
		return nil, nil
	}
	cids := make([]C.uint16_t, len(ids))
	for i, id := range ids {
		cids[i] = C.uint16_t(id)
	}
	vals := make([]C.gm_value_t, len(ids))
	if err := check("read fields", C.gm_smi_read_fields(C.uint32_t(idx), &cids[0], C.int(len(ids)), &vals[0])); err != nil {
		return nil, err
	}
	out := make([]interface{}, len(ids))
	for i := range vals {
		v := &vals[i]
		if v.status != C.GM_VS_OK {
			continue
		}
		switch v._type {
		case C.GM_FT_INT64:
			out[i] = int64(C.smi_value_i64(v))
		case C.GM_FT_DOUBLE:
			out[i] = float64(C.smi_value_dbl(v))
		case C.GM_FT_STRING:
			out[i] = C.GoString(C.smi_value_str(v))
		}
	}
	return out, nil
}

func fakeCtl(cmd string) error {
	c := C.CString(cmd)
	defer C.free(unsafe.Pointer(c))
	return check("fake ctl", C.gm_smi_fake_ctl(c))
}
