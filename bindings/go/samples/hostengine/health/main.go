// health runs every health watch on every GPU once a second and prints the
// result with its incidents (the reference's samples/dcgm/health).
package main

import (
	"fmt"
	"log"
	"os"
	"os/signal"
	"syscall"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

func main() {
	cleanup, err := he.Init(he.Embedded)
	if err != nil {
		log.Panicln("cannot start the host engine:", err)
	}
	defer cleanup()
	gpus, err := he.GetSupportedDevices()
	if err != nil {
		log.Panicln(err)
	}
	stop := make(chan os.Signal, 1)
	signal.Notify(stop, os.Interrupt, syscall.SIGTERM)
	tick := time.NewTicker(time.Second)
	defer tick.Stop()
	for {
		select {
		case <-tick.C:
			for _, g := range gpus {
				h, err := he.HealthCheckByGpuId(g)
				if err != nil {
					log.Panicf("GPU %d: %v", g, err)
				}
				fmt.Printf("GPU                : %d\nStatus             : %s\n", h.GPU, h.Status)
				for _, w := range h.Watches {
					fmt.Printf("Type               : %s\nStatus             : %s\nError              : %s\n", w.Type, w.Status, w.Error)
				}
				fmt.Println()
			}
		case <-stop:
			return
		}
	}
}
