// policy registers the critical-event policy (the XID analog: resets, VM faults,
// RAS errors) on every GPU and prints violations as they arrive (the reference's
// samples/dcgm/policy).  With -demo it runs on the fake backend and injects one.
package main

import (
	"flag"
	"fmt"
	"log"
	"os"
	"os/signal"
	"syscall"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

func main() {
	demo := flag.Bool("demo", false, "fake backend + one injected event, no GPU needed")
	flag.Parse()
	var cleanup func()
	var err error
	if *demo {
		cleanup, err = he.Init(he.Embedded, "fake", "gpus=2")
	} else {
		cleanup, err = he.Init(he.Embedded)
	}
	if err != nil {
		log.Panicln("cannot start the host engine:", err)
	}
	defer cleanup()
	gpus, err := he.GetSupportedDevices()
	if err != nil {
		log.Panicln(err)
	}
	merged := make(chan he.PolicyViolation, 16)
	for _, g := range gpus {
		ch, err := he.Policy(g, he.GpuEventPolicy, he.EccUePolicy, he.ThermalPolicy)
		if err != nil {
			log.Panicf("GPU %d: %v", g, err)
		}
		go func() {
			for v := range ch {
				merged <- v
			}
		}()
	}
	if *demo {
		go func() {
			time.Sleep(200 * time.Millisecond)
			_ = he.FakeCtl("event gpu=1 code=4 msg=injected-reset")
		}()
	}
	stop := make(chan os.Signal, 1)
	signal.Notify(stop, os.Interrupt, syscall.SIGTERM)
	log.Println("Policy successfully set; waiting for violations.")
	for {
		select {
		case v := <-merged:
			fmt.Printf("GPU %d  %v  %s  %+v\n", v.GPU, v.Timestamp.Format(time.RFC3339), v.Condition, v.Data)
			if *demo {
				return
			}
		case <-stop:
			return
		}
	}
}
