// deviceInfo prints identity, limits, clocks and peer topology of every GPU
// through the host engine; with -connect it talks to a running gpumon-hostengine
// instead of embedding one (the reference's samples/dcgm/deviceInfo).
package main

import (
	"flag"
	"fmt"
	"log"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

var (
	connect = flag.String("connect", "", "address of a running host engine: host:port, or a unix socket path with -socket")
	socket  = flag.Bool("socket", false, "the -connect address is a unix socket path")
)

func show(label string, v interface{}, unit string) {
	switch x := v.(type) {
	case *uint:
		if x != nil {
			fmt.Printf("%-23s: %d%s\n", label, *x, unit)
			return
		}
	case *int64:
		if x != nil {
			fmt.Printf("%-23s: %d%s\n", label, *x, unit)
			return
		}
	case string:
		if x != "" {
			fmt.Printf("%-23s: %s%s\n", label, x, unit)
			return
		}
	}
	fmt.Printf("%-23s: N/A\n", label)
}

func main() {
	flag.Parse()
	var cleanup func()
	var err error
	if *connect != "" {
		isSocket := "0"
		if *socket {
			isSocket = "1"
		}
		cleanup, err = he.Init(he.Standalone, *connect, isSocket)
	} else {
		cleanup, err = he.Init(he.Embedded)
	}
	if err != nil {
		log.Panicln("cannot reach the host engine:", err)
	}
	defer cleanup()
	count, err := he.GetAllDeviceCount()
	if err != nil {
		log.Panicln(err)
	}
	for i := uint(0); i < count; i++ {
		d, err := he.GetDeviceInfo(i)
		if err != nil {
			log.Panicf("GPU %d: %v", i, err)
		}
		supported := "No"
		if d.Supported {
			supported = "Yes"
		}
		fmt.Printf("%-23s: %d\n", "GPU", d.GPU)
		fmt.Printf("%-23s: %s\n", "Engine Supported", supported)
		show("Driver Version", d.Identifiers.DriverVersion, "")
		show("UUID", d.UUID, "")
		show("Brand", d.Identifiers.Brand, "")
		show("Model", d.Identifiers.Model, "")
		show("Serial Number", d.Identifiers.Serial, "")
		show("Vbios", d.Identifiers.Vbios, "")
		show("Gfx Target", d.Identifiers.GfxTarget, "")
		show("Bus ID", d.PCI.BusID, "")
		show("Visible VRAM", d.PCI.VisibleVRAM, " MiB")
		show("HBM Total", d.PCI.VRAMTotal, " MiB")
		show("PCIe Bandwidth", d.PCI.Bandwidth, " MB/s")
		show("Cores", d.Clocks.Cores, " MHz")
		show("Memory Clock", d.Clocks.Memory, " MHz")
		show("Power Cap", d.Power, " W")
		show("CPU Affinity", d.CPUAffinity, "")
		fmt.Printf("%-23s: %s / %s\n", "Partition", d.ComputePartition, d.MemoryPartition)
		fmt.Printf("%-23s: ", "P2P Available")
		if len(d.Topology) == 0 {
			fmt.Println("None")
		} else {
			fmt.Println()
			for _, p := range d.Topology {
				fmt.Printf("    GPU%d - (BusID)%s - %s\n", p.GPU, p.BusID, p.Name)
			}
		}
		fmt.Println("---------------------------------------------------------------------")
	}
}
