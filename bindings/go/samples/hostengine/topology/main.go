// topology prints the GPU-to-GPU link matrix with CPU affinity and a legend, in a
// private host engine started for the occasion (the reference's
// samples/dcgm/topology, which demonstrates the StartHostengine mode).
package main

import (
	"fmt"
	"log"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

const legend = `
Legend:
 X    = Self
 SYS  = Connection traversing PCIe as well as the interconnect between NUMA nodes
 NODE = Connection traversing PCIe and the interconnect between host bridges within a NUMA node
 PHB  = Connection traversing PCIe and a PCIe host bridge (typically the CPU)
 PXB  = Connection traversing multiple PCIe switches (without a host bridge)
 PIX  = Connection traversing a single PCIe switch
 PSB  = Connection traversing a single on-board PCIe switch
 XGMI# = Connection over xGMI (Infinity Fabric), # hops`

func main() {
	cleanup, err := he.Init(he.StartHostengine)
	if err != nil {
		log.Panicln("cannot start a host engine:", err)
	}
	defer cleanup()
	gpus, err := he.GetSupportedDevices()
	if err != nil {
		log.Panicln(err)
	}
	fmt.Printf("%-6s", "")
	for _, g := range gpus {
		fmt.Printf("%-7s", fmt.Sprintf("GPU%d", g))
	}
	fmt.Println("CPU Affinity")
	for _, g := range gpus {
		info, err := he.GetDeviceInfo(g)
		if err != nil {
			log.Panicf("GPU %d: %v", g, err)
		}
		row := map[uint]string{g: "X"}
		for _, p := range info.Topology {
			row[p.GPU] = p.Name
		}
		fmt.Printf("%-6s", fmt.Sprintf("GPU%d", g))
		for _, other := range gpus {
			name, ok := row[other]
			if !ok {
				name = "N/A"
			}
			fmt.Printf("%-7s", name)
		}
		fmt.Println(info.CPUAffinity)
	}
	fmt.Println(legend)
}
