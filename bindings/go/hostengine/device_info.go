package hostengine

/*
#include "capi.h"
*/
import "C"

// Same information as the reference's Device (bindings/go/dcgm/device_info.go:14-40),
// with AMD identifiers: the gfx target stands where the inforom version was, HBM
// where the framebuffer was, and partitioning is reported.

type PCIInfo struct {
	BusID       string
	VisibleVRAM *uint  // MiB, BAR1 analog
	VRAMTotal   *uint  // MiB of HBM
	Bandwidth   *int64 // MB/s, from the maximum link generation and width
}

type DeviceIdentifiers struct {
	Brand         string
	Model         string
	Serial        string
	Vbios         string
	Firmware      string
	DriverVersion string
	GfxTarget     string
}

type ClockInfo struct {
	Cores  *uint // MHz
	Memory *uint // MHz
}

type Device struct {
	GPU              uint
	Supported        bool // "DCGMSupported" in the reference: can the engine fully sample it
	UUID             string
	Path             string // /dev/dri/renderD<N>
	Power            *uint  // W, enforced cap
	DefaultPower     *uint  // W
	PCI              PCIInfo
	Clocks           ClockInfo // maximum clocks
	Identifiers      DeviceIdentifiers
	Topology         []P2PLink
	CPUAffinity      string // sysfs local_cpulist, e.g. "0-47,96-143"
	NumaNode         *uint
	NumXCD, NumCU    uint
	ComputePartition string
	MemoryPartition  string
}

func getAllDeviceCount() (uint, error) {
	var n C.uint32_t
	err := check("device count", C.gm_device_count(&n))
	return uint(n), err
}

func getSupportedDevices() ([]uint, error) {
	ids := make([]C.uint32_t, C.GM_MAX_GPUS)
	n := C.int(len(ids))
	if err := check("supported devices", C.gm_supported_devices(&ids[0], &n)); err != nil {
		return nil, err
	}
	out := make([]uint, int(n))
	for i := range out {
		out[i] = uint(ids[i])
	}
	return out, nil
}

func getDeviceInfo(gpuID uint) (Device, error) {
	var a C.gm_device_attributes_t
	C.he_attributes(&a)
	if err := check("device attributes", C.gm_device_attributes(C.uint32_t(gpuID), &a)); err != nil {
		return Device{}, err
	}
	d := Device{
		GPU: gpuID, Supported: a.supported != 0, UUID: C.GoString(&a.uuid[0]), Path: C.GoString(&a.device_path[0]),
		CPUAffinity: C.GoString(&a.cpu_affinity[0]), NumXCD: uint(a.num_xcd), NumCU: uint(a.num_cu),
		ComputePartition: C.GoString(&a.compute_partition[0]), MemoryPartition: C.GoString(&a.memory_partition[0]),
		Identifiers: DeviceIdentifiers{
			Brand: C.GoString(&a.brand[0]), Model: C.GoString(&a.name[0]), Serial: C.GoString(&a.serial[0]),
			Vbios: C.GoString(&a.vbios[0]), Firmware: C.GoString(&a.firmware[0]),
			DriverVersion: C.GoString(&a.driver_version[0]), GfxTarget: C.GoString(&a.gfx_target[0]),
		},
	}
	d.PCI.BusID = C.GoString(&a.pci_bdf[0])
	d.PCI.VisibleVRAM = optUint(a.vis_vram_total_mib, true)
	d.PCI.VRAMTotal = optUint(a.vram_total_mib, true)
	d.PCI.Bandwidth = optInt(a.pcie_bandwidth_mbs, a.pcie_bandwidth_mbs >= 0)
	d.Clocks = ClockInfo{Cores: optUint(a.gfx_clock_max_mhz, true), Memory: optUint(a.mem_clock_max_mhz, true)}
	if f := roundFloat(optFloat(a.power_cap_w, true)); f != nil {
		w := uint(*f)
		d.Power = &w
	}
	if f := roundFloat(optFloat(a.power_cap_default_w, true)); f != nil {
		w := uint(*f)
		d.DefaultPower = &w
	}
	d.NumaNode = optUint(C.int64_t(a.numa_node), a.numa_node >= 0)
	if d.Supported {
		topo, err := getDeviceTopology(gpuID)
		if err != nil {
			return d, err
		}
		d.Topology = topo
	}
	return d, nil
}
