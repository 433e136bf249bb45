// hostengineStatus prints the engine's own memory and CPU use (the reference's
// samples/dcgm/hostengineStatus) plus its sampler statistics.
package main

import (
	"fmt"
	"log"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

func main() {
	cleanup, err := he.Init(he.Embedded)
	if err != nil {
		log.Panicln("cannot start the host engine:", err)
	}
	defer cleanup()
	if _, err = he.Introspect(); err != nil { // enables the CPU accounting window
		log.Panicln(err)
	}
	time.Sleep(time.Second)
	st, err := he.Introspect()
	if err != nil {
		log.Panicln(err)
	}
	fmt.Printf("Memory(KB)      : %v\nCPU(%%)          : %.2f\nSample cycles   : %d\nAvg cycle       : %v\nBackend errors  : %d\n",
		st.Memory, st.CPU, st.SampleCycles, st.AvgCycle, st.BackendErrors)
}
