// dmon prints one line of telemetry per GPU per second through the direct device
// layer (the reference's samples/nvml/dmon).  Columns the hardware does not have
// (enc, dec on MI355X) print "-".
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

const header = `# gpu   pwr  temp    sm   mem   enc   dec  mclk  pclk
# Idx     W     C     %     %     %     %   MHz   MHz`

func cell(v *uint) string {
	if v == nil {
		return fmt.Sprintf("%5s", "-")
	}
	return fmt.Sprintf("%5d", *v)
}

func main() {
	if err := smi.Init(); err != nil {
		log.Panicln("cannot initialise the device layer:", err)
	}
	defer smi.Shutdown()
	count, err := smi.GetDeviceCount()
	if err != nil {
		log.Panicln(err)
	}
	devices := make([]*smi.Device, 0, count)
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
				st, err := d.Status()
				if err != nil {
					log.Panicf("GPU %d: %v", i, err)
				}
				fmt.Printf("%5d %s %s %s %s %s %s %s %s\n", i, cell(st.Power), cell(st.Temperature),
					cell(st.Utilization.GPU), cell(st.Utilization.Memory), cell(st.Utilization.Encoder),
					cell(st.Utilization.Decoder), cell(st.Clocks.Memory), cell(st.Clocks.Cores))
			}
		case <-stop:
			return
		}
	}
}
