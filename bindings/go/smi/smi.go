package smi

import (
	"errors"
	"fmt"
	"math"
	"strconv"
	"strings"
	"time"
)

// Public data model.  Same shape as the reference's nvml package
// (bindings/go/nvml/nvml.go:23-266): pointer members are optional and nil when the
// device does not report the value — never a sentinel number.

// ThrottleReason is a bit of DeviceStatus.Throttle.
type ThrottleReason uint

const (
	ThrottlePowerCap ThrottleReason = 1 << iota
	ThrottleThermal
	ThrottleHBMThermal
	ThrottleVRThermal
	ThrottleProcHot
	ThrottleLowUtilization
	ThrottleHostLimit
)

func (t ThrottleReason) String() string {
	if t == 0 {
		return "No clocks throttling"
	}
	names := []string{"Power cap", "Thermal", "HBM thermal", "VR thermal", "PROCHOT", "Low utilization", "Host limit"}
	var on []string
	for i, n := range names {
		if t&(1<<uint(i)) != 0 {
			on = append(on, n)
		}
	}
	return strings.Join(on, ", ")
}

// P2PLinkType classifies the path between two GPUs.
type P2PLinkType uint

func (l P2PLinkType) String() string { return linkDescription(uint(l)) }

// P2PLink is one peer of a GPU: the PEER's bus id and how it is reached ("XGMI1", "PIX", "SYS"...).
type P2PLink struct {
	BusID string
	Link  P2PLinkType
	Name  string
}

type ClockInfo struct {
	Cores  *uint // MHz
	Memory *uint // MHz
}

type PCIInfo struct {
	BusID       string
	VisibleVRAM *uint64 // MiB; the BAR1 analog (all of HBM on MI355X: large BAR)
	Bandwidth   *uint   // MB/s, from the maximum link generation and width
}

// Device is the static identity of one GPU.
type Device struct {
	Index            uint
	UUID             string
	Path             string // /dev/dri/renderD<N>
	Model            *string
	Power            *uint   // W, enforced power cap
	Memory           *uint64 // MiB of HBM
	CPUAffinity      *uint   // NUMA node
	PCI              PCIInfo
	Clocks           ClockInfo // maximum clocks
	Topology         []P2PLink
	GfxTarget        string // e.g. gfx950; the compute-capability analog
	NumXCD, NumCU    uint
	ComputePartition string // SPX, DPX, QPX, CPX
	MemoryPartition  string // NPS1, NPS2, ...
	Serial, VBIOS    string
}

type UtilizationInfo struct {
	GPU     *uint // %
	Memory  *uint // % (HBM controller activity)
	Encoder *uint // nil on MI355X: there is no video encoder
	Decoder *uint
}

type ECCErrorsInfo struct {
	Correctable   *uint64
	Uncorrectable *uint64
}

// DeviceMemory and MemoryInfo nest like the reference's (nvml.go:239-247): st.Memory.Global.Used.
type DeviceMemory struct {
	Used *uint64 // MiB of HBM
	Free *uint64 // MiB
}

type MemoryInfo struct {
	Global    DeviceMemory
	ECCErrors ECCErrorsInfo
}

type PCIThroughputInfo struct {
	RX *uint // MB/s; nil where the device library has no PCIe traffic counters
	TX *uint
}

// PCIStatusInfo nests like the reference's (nvml.go:228-231): st.PCI.Throughput.RX.  VisibleVRAMUsed is
// the BAR1Used analog: MiB in use of the CPU-visible aperture (all of HBM on a large-BAR MI355X).
type PCIStatusInfo struct {
	VisibleVRAMUsed *uint64
	Throughput      PCIThroughputInfo
	Replays         *uint64 // PCIe replay counter
}

// PerfState is the driver's performance level, the P-state analog (nvml.go:98-111).  The amdgpu driver has
// named levels instead of P0..P15.
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

// ProcessType keeps the reference's three kinds (nvml.go:113-131).  KFD processes are compute
// processes; an MI355X has no graphics queue, so Graphics never occurs on it.
type ProcessType uint

const (
	Compute ProcessType = iota
	Graphics
	ComputeAndGraphics
)

func (t ProcessType) String() string {
	switch t {
	case Compute:
		return "C"
	case Graphics:
		return "G"
	}
	return "C+G"
}

type ProcessInfo struct {
	PID         uint
	Name        string
	MemoryUsed  uint64 // MiB; 0 when MemoryKnown is false
	MemoryKnown bool   // false: the process holds the device open, the kernel does not report its memory
	Type        ProcessType
	CUOccupancy *uint
}

// DeviceStatus is one telemetry read.
type DeviceStatus struct {
	Power       *uint // W
	Temperature *uint // °C, hotspot
	HBMTemp     *uint // °C
	Utilization UtilizationInfo
	Memory      MemoryInfo
	Clocks      ClockInfo
	PCI         PCIStatusInfo
	Processes   []ProcessInfo
	Throttle    *ThrottleReason
	Performance *PerfState
	EnergyJ     *float64 // energy counter since driver load
}

// DeviceMode keeps the reference's three members (nvml.go:52-56, 582-604) and adds what a
// compute-only part has instead of display and persistence modes: its partitioning.
type DeviceMode struct {
	DisplayInfo      Display    // always N/A: no display engine on the part
	Persistence      ModeState  // always N/A: the amdgpu driver stays loaded, there is no such mode
	AccountingInfo   Accounting // the engine's per-process accounting watch on this GPU
	ComputePartition string     // SPX, DPX, QPX, CPX
	MemoryPartition  string     // NPS1, NPS2, ...
}

// ModeState, Display and Accounting keep the reference's shapes (nvml.go:23-50).
type ModeState uint

const (
	Disabled ModeState = iota
	Enabled
	NotApplicable
)

func (m ModeState) String() string {
	switch m {
	case Enabled:
		return "Enabled"
	case Disabled:
		return "Disabled"
	}
	return "N/A"
}

type Display struct {
	Mode   ModeState
	Active ModeState
}

type Accounting struct {
	Mode       ModeState
	BufferSize *uint
}

// Event is a driver notification (GPU reset, RAS error, thermal throttle ...): the
// counterpart of the reference's XID events (bindings.go:68-146).
type Event struct {
	GPU     uint
	UUID    string
	Code    uint // GM_EVENT_* code
	Name    string
	Message string
	Time    time.Time
	Etype   uint64 // 1 << Code: the bit RegisterEvent takes (the reference's Event.Etype)
	Edata   uint64 // the code again (the reference's Event.Edata carries the XID number)
}

var ErrUnsupportedGPU = errors.New("unsupported GPU device")

// Init loads the device library and enumerates GPUs.  Reference-counted.
func Init() error { return initLib("", "") }

// InitWithBackend selects a backend explicitly ("amdsmi" or "fake") — tests use the fake.
func InitWithBackend(backend, spec string) error { return initLib(backend, spec) }

func Shutdown() error { return shutdownLib() }

func GetDeviceCount() (uint, error) { return deviceCount() }

func GetDriverVersion() (string, error) { return driverVersion() }

// GetRuntimeVersion returns major and minor of the ROCm release the device library belongs to, the
// counterpart of nvml.GetCudaDriverVersion (bindings/go/nvml/nvml.go:290-292); nil, nil when unknown.
func GetRuntimeVersion() (*uint, *uint, error) {
	major, minor, _, err := runtimeVersion()
	return major, minor, err
}

// GfxVersionInfo is the compute-capability analog (bindings/go/nvml/nvml.go:196-199): gfx950 is
// {9, 5, 0}, gfx90a is {9, 0, 10}.  Nil members when the target string is not of that form.
type GfxVersionInfo struct {
	Major, Minor, Stepping *int
}

// GfxVersion parses Device.GfxTarget.
func (d *Device) GfxVersion() GfxVersionInfo {
	t := d.GfxTarget
	if i := strings.IndexByte(t, ':'); i >= 0 {
		t = t[:i]
	}
	if !strings.HasPrefix(t, "gfx") || len(t) < 6 {
		return GfxVersionInfo{}
	}
	major, e1 := strconv.ParseInt(t[3:len(t)-2], 10, 32)
	minor, e2 := strconv.ParseInt(t[len(t)-2:len(t)-1], 16, 32)
	step, e3 := strconv.ParseInt(t[len(t)-1:], 16, 32)
	if e1 != nil || e2 != nil || e3 != nil {
		return GfxVersionInfo{}
	}
	ma, mi, st := int(major), int(minor), int(step)
	return GfxVersionInfo{&ma, &mi, &st}
}

func optUint(v int64, ok bool) *uint {
	if !ok || v < 0 {
		return nil
	}
	u := uint(v)
	return &u
}

func optU64(v int64, ok bool) *uint64 {
	if !ok || v < 0 {
		return nil
	}
	u := uint64(v)
	return &u
}

func newDevice(idx uint, lite bool) (*Device, error) {
	a, err := deviceAttributes(idx)
	if err != nil {
		return nil, err
	}
	if !a.supported || a.uuid == "" {
		return nil, ErrUnsupportedGPU
	}
	d := &Device{Index: idx, UUID: a.uuid, Path: a.path, GfxTarget: a.gfxTarget, NumXCD: a.numXCD, NumCU: a.numCU,
		ComputePartition: a.computePart, MemoryPartition: a.memPart, Serial: a.serial, VBIOS: a.vbios}
	d.PCI.BusID = a.busID
	d.CPUAffinity = optUint(int64(a.numaNode), a.numaNode >= 0)
	if a.name != "" {
		name := a.name
		d.Model = &name
	}
	if lite {
		return d, nil
	}
	if !math.IsNaN(a.powerCapW) {
		d.Power = optUint(int64(math.Round(a.powerCapW)), true)
	}
	d.Memory = optU64(a.vramTotalMiB, true)
	d.PCI.VisibleVRAM = optU64(a.visVramTotalMiB, true)
	d.PCI.Bandwidth = optUint(a.pcieBandwidthMBs, true)
	d.Clocks.Cores = optUint(a.gfxClockMaxMHz, true)
	d.Clocks.Memory = optUint(a.memClockMaxMHz, true)
	n, err := deviceCount()
	if err != nil {
		return nil, err
	}
	for peer := uint(0); peer < n; peer++ {
		if peer == idx {
			continue
		}
		l, err := peerLink(idx, peer)
		if isStatus(err, StatusNotSupported) {
			d.Topology = nil
			break
		}
		if err != nil {
			return nil, err
		}
		d.Topology = append(d.Topology, P2PLink{BusID: l.busID, Link: P2PLinkType(l.link), Name: linkName(l.link, l.hops)})
	}
	return d, nil
}

// NewDevice reads the full static description of GPU idx, including its topology.
func NewDevice(idx uint) (*Device, error) { return newDevice(idx, false) }

// NewDeviceLite reads only identity (UUID, path, bus id, model, NUMA node): what a
// device plugin needs at start-up.
func NewDeviceLite(idx uint) (*Device, error) { return newDevice(idx, true) }

// Status performs one synchronous telemetry read.
func (d *Device) Status() (*DeviceStatus, error) {
	s, err := deviceStatus(d.Index)
	if err != nil {
		return nil, err
	}
	ok := func(bit uint) bool { return s.valid&(1<<bit) != 0 }
	st := &DeviceStatus{}
	if ok(0) {
		st.Power = optUint(int64(s.powerW), true) // truncated, like the reference's mW -> W
	}
	st.Temperature = optUint(s.temp, ok(1))
	st.Utilization = UtilizationInfo{GPU: optUint(s.gfxUtil, ok(2)), Memory: optUint(s.memUtil, ok(3)),
		Encoder: optUint(s.encUtil, ok(4)), Decoder: optUint(s.decUtil, ok(5))}
	st.Clocks = ClockInfo{Cores: optUint(s.gfxClock, ok(6)), Memory: optUint(s.memClock, ok(7))}
	st.PCI = PCIStatusInfo{VisibleVRAMUsed: optU64(s.visVramUsed, ok(8)), Replays: optU64(s.pcieReplays, ok(11)),
		Throughput: PCIThroughputInfo{RX: optUint(s.pcieRxKB/1000, ok(9)), TX: optUint(s.pcieTxKB/1000, ok(10))}}
	st.Memory = MemoryInfo{Global: DeviceMemory{Used: optU64(s.vramUsed, ok(12)), Free: optU64(s.vramFree, ok(13))},
		ECCErrors: ECCErrorsInfo{Correctable: optU64(s.eccCE, ok(14)), Uncorrectable: optU64(s.eccUE, ok(15))}}
	if ok(16) {
		ps := PerfState(s.perfLevel)
		st.Performance = &ps
	}
	if ok(18) {
		t := ThrottleReason(s.throttle)
		st.Throttle = &t
	}
	st.HBMTemp = optUint(s.memTemp, ok(19))
	if ok(20) {
		j := float64(s.energyMilliJ) / 1000.0
		st.EnergyJ = &j
	}
	procs, err := d.GetAllRunningProcesses()
	if err != nil && !isStatus(err, StatusNotSupported) && !isStatus(err, StatusNoPermission) {
		return nil, err
	}
	st.Processes = procs
	return st, nil
}

// GetAllRunningProcesses lists the processes that hold the GPU.
func (d *Device) GetAllRunningProcesses() ([]ProcessInfo, error) {
	raw, err := deviceProcesses(d.Index)
	if err != nil {
		return nil, err
	}
	out := make([]ProcessInfo, 0, len(raw))
	for _, p := range raw {
		pi := ProcessInfo{PID: p.pid, Name: p.name, MemoryUsed: p.vramBytes >> 20, MemoryKnown: true}
		if p.vramBytes == procMemUnknown {
			pi.MemoryUsed, pi.MemoryKnown = 0, false
		}
		if p.cuOccupancy != 0xffffffff {
			cu := p.cuOccupancy
			pi.CUOccupancy = &cu
		}
		out = append(out, pi)
	}
	return out, nil
}

// GetP2PLink classifies the path between two devices.
func GetP2PLink(dev1, dev2 *Device) (P2PLinkType, string, error) {
	l, err := peerLink(dev1.Index, dev2.Index)
	if err != nil {
		return 0, "", err
	}
	return P2PLinkType(l.link), linkName(l.link, l.hops), nil
}

// GetXGMILink reports whether two devices are directly connected over xGMI and
// with how many links; the counterpart of GetNVLink (nvml.go:539-568).
func GetXGMILink(dev1, dev2 *Device) (connected bool, links uint, err error) {
	l, err := peerLink(dev1.Index, dev2.Index)
	if err != nil {
		return false, 0, err
	}
	name := linkName(l.link, l.hops)
	return strings.HasPrefix(name, "XGMI"), l.numLinks, nil
}

// XgmiLink is one physical link of a device, as the reference's per-link walk over
// NvLinkState / NvLinkRemotePciInfo reports it (bindings/go/nvml/bindings.go:268-318).
type XgmiLink struct {
	Link       uint
	Up         *bool   // nil: not wired or not reported
	PeerIndex  *uint   // device index at the far end
	PeerBusID  string  // empty when unknown
	BitRateGbs *uint   // per lane
	MaxBwGbs   *uint   // advertised, one direction
	RxKB, TxKB *uint64 // cumulative
}

// GetXGMILinks lists every link of the device with its far end.
func (d *Device) GetXGMILinks() ([]XgmiLink, error) {
	raw, err := xgmiLinks(d.Index)
	if err != nil {
		return nil, err
	}
	out := make([]XgmiLink, 0, len(raw))
	for _, l := range raw {
		x := XgmiLink{Link: l.link, PeerIndex: optUint(int64(l.peer), l.peer >= 0), PeerBusID: l.peerBusID,
			BitRateGbs: optUint(l.bitRate, l.bitRate >= 0), MaxBwGbs: optUint(l.maxBw, l.maxBw >= 0),
			RxKB: optU64(l.rxKB, l.rxKB >= 0), TxKB: optU64(l.txKB, l.txKB >= 0)}
		if l.state >= 0 {
			up := l.state != 0
			x.Up = &up
		}
		out = append(out, x)
	}
	return out, nil
}

func (d *Device) GetDeviceMode() (*DeviceMode, error) {
	m, err := deviceMode(d.Index)
	if err != nil {
		return nil, err
	}
	mode := &DeviceMode{DisplayInfo: Display{NotApplicable, NotApplicable}, Persistence: NotApplicable,
		ComputePartition: m.compute, MemoryPartition: m.memory}
	if m.accountingOn {
		size := m.accountingBufferSize
		mode.AccountingInfo = Accounting{Mode: Enabled, BufferSize: &size}
	}
	return mode, nil
}

// GetComputeRunningProcesses returns pids and used memory in MiB (nvml.go:570-572).
func (d *Device) GetComputeRunningProcesses() ([]uint, []uint64, error) {
	procs, err := d.GetAllRunningProcesses()
	if err != nil {
		return nil, nil, err
	}
	pids, mems := make([]uint, 0, len(procs)), make([]uint64, 0, len(procs))
	for _, p := range procs {
		if p.Type != Graphics {
			pids, mems = append(pids, p.PID), append(mems, p.MemoryUsed)
		}
	}
	return pids, mems, nil
}

// GetGraphicsRunningProcesses is always empty on a compute-only part (nvml.go:574-576).
func (d *Device) GetGraphicsRunningProcesses() ([]uint, []uint64, error) {
	return []uint{}, []uint64{}, nil
}

// ReadFields reads arbitrary registry fields (ids from the C header, GM_FI_*) in
// one round; entries are int64, float64, string or nil.
func (d *Device) ReadFields(ids []uint16) ([]interface{}, error) { return readFields(d.Index, ids) }

// WaitForAnyEvent blocks up to timeout for the next driver event of any type on any GPU;
// (nil, nil) on timeout.  WaitForEvent (events.go) is the filtered, reference-shaped form.
func WaitForAnyEvent(timeout time.Duration) (*Event, error) {
	e, err := eventWait(int(timeout / time.Millisecond))
	if isStatus(err, StatusTimeout) {
		return nil, nil
	}
	if err != nil {
		return nil, err
	}
	ev := makeEvent(e)
	return &ev, nil
}

func (d *Device) String() string {
	model := "unknown"
	if d.Model != nil {
		model = *d.Model
	}
	return fmt.Sprintf("GPU %d: %s (%s) %s", d.Index, model, d.UUID, d.PCI.BusID)
}
