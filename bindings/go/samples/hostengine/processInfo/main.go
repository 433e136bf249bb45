// processInfo starts per-process accounting, waits for a few samples and prints
// what was accounted to one PID on every GPU it used (the reference's
// samples/dcgm/processInfo; its SM / memory utilisation labels are swapped, ours are not).
package main

import (
	"flag"
	"fmt"
	"log"
	"time"

	he "github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/hostengine"
)

func f(v *float64, unit string) string {
	if v == nil {
		return "N/A"
	}
	return fmt.Sprintf("%.1f%s", *v, unit)
}

func i(v *int64, unit string) string {
	if v == nil {
		return "N/A"
	}
	return fmt.Sprintf("%d%s", *v, unit)
}

func u(v *uint, unit string) string {
	if v == nil {
		return "N/A"
	}
	return fmt.Sprintf("%d%s", *v, unit)
}

func d(v *time.Duration) string {
	if v == nil {
		return "N/A"
	}
	return v.String()
}

func main() {
	pid := flag.Uint("pid", 0, "process id to report on")
	wait := flag.Duration("wait", 3*time.Second, "accounting warm-up before the read")
	flag.Parse()
	cleanup, err := he.Init(he.Embedded)
	if err != nil {
		log.Panicln("cannot start the host engine:", err)
	}
	defer cleanup()
	group, err := he.WatchPidFields()
	if err != nil {
		log.Panicln(err)
	}
	log.Printf("Enabled per-process accounting; sampling for %v ...", *wait)
	time.Sleep(*wait)
	infos, err := he.GetProcessInfo(group, *pid)
	if err != nil {
		log.Panicf("pid %d: %v", *pid, err)
	}
	for _, p := range infos {
		fmt.Printf("----------------------------------------------------------------------\n")
		fmt.Printf("GPU ID                       : %d\n", p.GPU)
		fmt.Printf("----------Execution Stats---------------------------------------------\n")
		fmt.Printf("PID                          : %d\nName                         : %s\n", p.PID, p.Name)
		fmt.Printf("Start Time                   : %s\nEnd Time                     : %s\n", p.ProcessUtilization.StartTime, p.ProcessUtilization.EndTime)
		fmt.Printf("----------Performance Stats-------------------------------------------\n")
		fmt.Printf("Energy Consumed              : %s\n", f(p.ProcessUtilization.EnergyConsumed, " J"))
		fmt.Printf("Max HBM Used                 : %s\n", i(p.Memory.GlobalUsed, " B"))
		fmt.Printf("Avg Gfx Clock                : %s\nAvg Memory Clock             : %s\n", u(p.Clocks.Cores, " MHz"), u(p.Clocks.Memory, " MHz"))
		fmt.Printf("Avg Gfx Utilization          : %s\nAvg Memory Utilization       : %s\n", u(p.GpuUtilization.GPU, " %"), u(p.GpuUtilization.Memory, " %"))
		fmt.Printf("Process Gfx Share            : %s\n", f(p.ProcessUtilization.GfxUtil, " %"))
		fmt.Printf("Avg PCIe Rx / Tx             : %s / %s\n", i(p.PCI.AvgRxKBs, " KB/s"), i(p.PCI.AvgTxKBs, " KB/s"))
		fmt.Printf("----------Event Stats-------------------------------------------------\n")
		fmt.Printf("ECC Correctable / Uncorr.    : %s / %s\n", u(p.Memory.ECCErrors.Correctable, ""), u(p.Memory.ECCErrors.Uncorrectable, ""))
		fmt.Printf("Critical Events              : %d %v\n", p.XIDErrors.NumErrors, p.XIDErrors.Timestamp)
		fmt.Printf("----------Slowdown Stats----------------------------------------------\n")
		fmt.Printf("Power / Thermal              : %s / %s\n", d(p.Violations.Power), d(p.Violations.Thermal))
		fmt.Printf("Reliability / Board Limit    : %s / %s\n", d(p.Violations.Reliability), d(p.Violations.BoardLimit))
		fmt.Printf("Low Utilization              : %s\n", d(p.Violations.LowUtilization))
	}
}
