// deviceInfo prints the static description of every GPU through the direct
// device layer (the reference's samples/nvml/deviceInfo, `nvidia-smi -q` style).
package main

import (
	"fmt"
	"log"
	"os"
	"text/template"

	"github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/smi"
)

const report = `UUID           : {{.UUID}}
Model          : {{or .Model "N/A"}}
Path           : {{.Path}}
Gfx Target     : {{.GfxTarget}}
XCDs / CUs     : {{.NumXCD}} / {{.NumCU}}
Partition      : {{.ComputePartition}} / {{.MemoryPartition}}
Power          : {{if .Power}}{{.Power}} W{{else}}N/A{{end}}
Memory         : {{if .Memory}}{{.Memory}} MiB{{else}}N/A{{end}}
CPU Affinity   : {{if .CPUAffinity}}NUMA node{{.CPUAffinity}}{{else}}N/A{{end}}
Bus ID         : {{.PCI.BusID}}
Visible VRAM   : {{if .PCI.VisibleVRAM}}{{.PCI.VisibleVRAM}} MiB{{else}}N/A{{end}}
Bandwidth      : {{if .PCI.Bandwidth}}{{.PCI.Bandwidth}} MB/s{{else}}N/A{{end}}
Cores          : {{if .Clocks.Cores}}{{.Clocks.Cores}} MHz{{else}}N/A{{end}}
Memory Clock   : {{if .Clocks.Memory}}{{.Clocks.Memory}} MHz{{else}}N/A{{end}}
P2P Available  : {{if not .Topology}}None{{else}}{{range .Topology}}
    {{.BusID}} - {{.Name}}{{end}}{{end}}
---------------------------------------------------------------------
`

func main() {
	if err := smi.Init(); err != nil {
		log.Panicln("cannot initialise the device layer:", err)
	}
	defer smi.Shutdown()
	count, err := smi.GetDeviceCount()
	if err != nil {
		log.Panicln("cannot count devices:", err)
	}
	if v, err := smi.GetDriverVersion(); err == nil {
		fmt.Printf("Driver Version : %s\n", v)
	}
	// Optional members are pointers; deref them for printing.
	t := template.Must(template.New("device").Funcs(template.FuncMap{}).Parse(report))
	for i := uint(0); i < count; i++ {
		d, err := smi.NewDevice(i)
		if err != nil {
			log.Panicf("GPU %d: %v", i, err)
		}
		fmt.Printf("GPU %12s %d\n", ":", i)
		if err = t.Execute(os.Stdout, deref(d)); err != nil {
			log.Panicln("template:", err)
		}
	}
}

// deref flattens the optional pointers so the template prints values, not addresses.
func deref(d *smi.Device) map[string]interface{} {
	opt := func(p interface{}) interface{} {
		switch v := p.(type) {
		case *uint:
			if v != nil {
				return *v
			}
		case *uint64:
			if v != nil {
				return *v
			}
		case *string:
			if v != nil {
				return *v
			}
		}
		return nil
	}
	return map[string]interface{}{
		"UUID": d.UUID, "Model": opt(d.Model), "Path": d.Path, "GfxTarget": d.GfxTarget, "NumXCD": d.NumXCD,
		"NumCU": d.NumCU, "ComputePartition": d.ComputePartition, "MemoryPartition": d.MemoryPartition,
		"Power": opt(d.Power), "Memory": opt(d.Memory), "CPUAffinity": opt(d.CPUAffinity),
		"PCI":    map[string]interface{}{"BusID": d.PCI.BusID, "VisibleVRAM": opt(d.PCI.VisibleVRAM), "Bandwidth": opt(d.PCI.Bandwidth)},
		"Clocks": map[string]interface{}{"Cores": opt(d.Clocks.Cores), "Memory": opt(d.Clocks.Memory)},
		"Topology": d.Topology,
	}
}
