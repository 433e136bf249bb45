// dmon prints one line of cached telemetry per GPU per second from an embedded
// host engine (the reference's samples/dcgm/dmon).
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

const header = `# gpu   pwr  temp    sm   mem   enc   dec  mclk  pclk
# Idx     W     C     %     %     %     %   MHz   MHz`

func cell(v *uint) string {
	if v == nil {
		return fmt.Sprintf("%5s", "-")
	}
	return fmt.Sprintf("%5d", *v)
}

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
	fmt.Println(header)
	for {
		select {
		case <-tick.C:
			for _, g := range gpus {
				st, err := he.GetDeviceStatus(g)
				if err != nil {
					log.Panicf("GPU %d: %v", g, err)
				}
				pwr := fmt.Sprintf("%5s", "-")
				if st.Power != nil {
					pwr = fmt.Sprintf("%5.0f", *st.Power)
				}
				fmt.Printf("%5d %s %s %s %s %s %s %s %s\n", g, pwr, cell(st.Temperature), cell(st.Utilization.GPU),
					cell(st.Utilization.Memory), cell(st.Utilization.Encoder), cell(st.Utilization.Decoder),
					cell(st.Clocks.Memory), cell(st.Clocks.Cores))
			}
		case <-stop:
			return
		}
	}
}
