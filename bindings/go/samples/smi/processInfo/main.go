// processInfo lists, once a second, the processes holding each GPU (the
// reference's samples/nvml/processInfo).  MI355X runs compute queues only, so the
// type column is always C.
package main

import (
	"fmt"
	"log"
	"os"
	"os/signal"
	"syscall"
	"time"

	"github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/smi"
)

const header = `# gpu   pid   type  mem  cus  Command
# Idx     #   C/G   MiB    #  name`

func main() {
	if err := smi.Init(); err != nil {
		log.Panicln("cannot initialise the device layer:", err)
	}
	defer smi.Shutdown()
	count, err := smi.GetDeviceCount()
	if err != nil {
		log.Panicln(err)
	}
	var devices []*smi.Device
	for i := uint(0); i < count; i++ {
		d, err := smi.NewDeviceLite(i)
		if err != nil {
			log.Panicf("GPU %d: %v", i, err)
		}
		devices = append(devices, d)
	}
	stop := make(chan os.Signal, 1)
	signal.Notify(stop, os.Interrupt, syscall.SIGTERM)
	tick := time.NewTicker(time.Second)
	defer tick.Stop()
	fmt.Println(header)
	for {
		select {
		case <-tick.C:
			for i, d := range devices {
				procs, err := d.GetAllRunningProcesses()
				if err != nil {
					log.Panicf("GPU %d: %v", i, err)
				}
				if len(procs) == 0 {
					fmt.Printf("%5d %5s %5s %5s %4s  %s\n", i, "-", "-", "-", "-", "-")
				}
				for _, p := range procs {
					cus := "-"
					if p.CUOccupancy != nil {
						cus = fmt.Sprint(*p.CUOccupancy)
					}
					fmt.Printf("%5d %5d %5s %5d %4s  %s\n", i, p.PID, p.Type, p.MemoryUsed, cus, p.Name)
				}
			}
		case <-stop:
			return
		}
	}
}
