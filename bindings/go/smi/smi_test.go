package smi

// Two kinds of tests.  The fake-backend tests run anywhere (the reference has
// nothing like them).  The hardware tests compare with `amd-smi`, as the
// reference's tests compare with nvidia-smi (bindings/go/nvml/nvml_test.go:18-218),
// with tolerances for instantaneous telemetry instead of exact equality; they are
// skipped when there is no GPU.  Benchmarks mirror BenchmarkDeviceCount1 /
// BenchmarkDeviceInfo1 (nvml_test.go:33-43,118-129).

import (
	"strings"
	"testing"
	"time"

	"github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/smi/amdsmicli"
)

func withFake(t testing.TB, spec string) func() {
	t.Helper()
	if err := InitWithBackend("fake", spec); err != nil {
		t.Fatalf("init: %v", err)
	}
	return func() {
		if err := Shutdown(); err != nil {
			t.Fatalf("shutdown: %v", err)
		}
	}
}

func TestFakeDeviceCountAndInfo(t *testing.T) {
	defer withFake(t, "gpus=4,seed=5,steady=1")()
	n, err := GetDeviceCount()
	if err != nil || n != 4 {
		t.Fatalf("count = %d, %v", n, err)
	}
	d, err := NewDevice(2)
	if err != nil {
		t.Fatal(err)
	}
	if d.Model == nil || *d.Model != "AMD Instinct MI355X" || !strings.HasPrefix(d.Path, "/dev/dri/renderD") {
		t.Errorf("unexpected identity: %v", d)
	}
	if d.Memory == nil || *d.Memory != 294896 || d.Power == nil || *d.Power != 1400 {
		t.Errorf("memory/power: %v %v", d.Memory, d.Power)
	}
	if len(d.Topology) != 3 || d.Topology[0].Name != "XGMI1" {
		t.Errorf("topology: %+v", d.Topology)
	}
	if v := d.GfxVersion(); v.Major == nil || *v.Major != 9 || *v.Minor != 5 || *v.Stepping != 0 {
		t.Errorf("gfx version of %q: %+v", d.GfxTarget, v)
	}
	if v := (&Device{GfxTarget: "gfx90a:sramecc+:xnack-"}).GfxVersion(); v.Major == nil || *v.Major != 9 || *v.Minor != 0 || *v.Stepping != 10 {
		t.Errorf("gfx90a: %+v", v)
	}
	if major, minor, err := GetRuntimeVersion(); err != nil || major == nil || *major != 7 || *minor != 2 {
		t.Errorf("runtime version: %v %v %v", major, minor, err)
	}
	lite, err := NewDeviceLite(2)
	if err != nil || lite.UUID != d.UUID || lite.Power != nil || lite.Topology != nil {
		t.Errorf("lite: %+v %v", lite, err)
	}
	if _, err := NewDevice(9); err == nil {
		t.Error("out-of-range index accepted")
	}
}

func TestFakeStatusOptionalValues(t *testing.T) {
	defer withFake(t, "gpus=2,seed=5,steady=1")()
	d, _ := NewDeviceLite(1)
	st, err := d.Status()
	if err != nil {
		t.Fatal(err)
	}
	if st.Utilization.Encoder != nil || st.Utilization.Decoder != nil {
		t.Error("MI355X has no encoder: the value must be nil, not 0")
	}
	if st.Temperature == nil || st.Power == nil || st.Clocks.Cores == nil || st.Throttle == nil {
		t.Fatalf("missing telemetry: %+v", st)
	}
	if st.Throttle.String() != "No clocks throttling" {
		t.Errorf("throttle: %v", st.Throttle)
	}
	if err := fakeCtl("unsupported gpu=1 field=155"); err != nil {
		t.Fatal(err)
	}
	if st, _ = d.Status(); st.Power != nil {
		t.Error("NOT_SUPPORTED must surface as nil")
	}
	if err := fakeCtl("set gpu=1 field=150 value=93"); err != nil {
		t.Fatal(err)
	}
	if st, _ = d.Status(); st.Temperature == nil || *st.Temperature != 93 {
		t.Errorf("no cache in the direct layer: want 93, got %v", st.Temperature)
	}
}

func TestFakeEventsAndProcesses(t *testing.T) {
	defer withFake(t, "gpus=2")()
	if e, err := WaitForAnyEvent(50 * time.Millisecond); e != nil || err != nil {
		t.Fatalf("expected a timeout, got %v %v", e, err)
	}
	go func() {
		time.Sleep(50 * time.Millisecond)
		_ = fakeCtl("event gpu=1 code=5 msg=ras")
	}()
	e, err := WaitForAnyEvent(3 * time.Second)
	if err != nil || e == nil || e.GPU != 1 || e.Code != 5 {
		t.Fatalf("event: %+v %v", e, err)
	}
	// The reference's shape: an event set registered for critical errors of one device.
	d1, _ := NewDeviceLite(1)
	es := NewEventSet()
	defer DeleteEventSet(es)
	if err := RegisterEventForDevice(es, XidCriticalError, "no-such-uuid"); err == nil {
		t.Fatal("an unknown UUID must be refused")
	}
	if err := RegisterEventForDevice(es, XidCriticalError, d1.UUID); err != nil {
		t.Fatal(err)
	}
	if _, err := WaitForEvent(es, 50); !isStatus(err, StatusTimeout) {
		t.Fatalf("expected a timeout status, got %v", err)
	}
	go func() {
		time.Sleep(50 * time.Millisecond)
		_ = fakeCtl("event gpu=0 code=5 msg=other-gpu") // filtered out: not registered for GPU 0
		_ = fakeCtl("event gpu=1 code=2 msg=thermal")   // filtered out: not a critical error
		_ = fakeCtl("event gpu=1 code=1 msg=vmfault")
	}()
	ev, err := WaitForEvent(es, 3000)
	if err != nil || ev.UUID != d1.UUID || ev.Etype != EventVMFault || ev.Edata != 1 {
		t.Fatalf("event set: %+v %v", ev, err)
	}
	_ = fakeCtl("proc_add gpu=0 pid=77 name=job vram=2147483648")
	d, _ := NewDeviceLite(0)
	p, err := d.GetAllRunningProcesses()
	if err != nil || len(p) != 1 || p[0].PID != 77 || p[0].MemoryUsed != 2048 {
		t.Fatalf("processes: %+v %v", p, err)
	}
	if p[0].Type != Compute || p[0].Type.String() != "C" {
		t.Errorf("process type: %v", p[0].Type)
	}
	pids, mems, err := d.GetComputeRunningProcesses()
	if err != nil || len(pids) != 1 || pids[0] != 77 || mems[0] != 2048 {
		t.Errorf("compute processes: %v %v %v", pids, mems, err)
	}
	if gp, gm, err := d.GetGraphicsRunningProcesses(); err != nil || len(gp) != 0 || len(gm) != 0 {
		t.Errorf("graphics processes on a compute-only part: %v %v %v", gp, gm, err)
	}
	m, err := d.GetDeviceMode()
	if err != nil || m.ComputePartition != "SPX" || m.DisplayInfo.Mode.String() != "N/A" || m.Persistence.String() != "N/A" ||
		m.AccountingInfo.Mode != Disabled || m.AccountingInfo.BufferSize != nil {
		t.Errorf("device mode: %+v %v", m, err)
	}
}

func TestInitRefcount(t *testing.T) {
	if err := Shutdown(); err == nil {
		t.Error("shutdown without init must fail")
	}
	stop1 := withFake(t, "gpus=2")
	stop2 := withFake(t, "gpus=6") // only takes a reference
	if n, _ := GetDeviceCount(); n != 2 {
		t.Errorf("second init must not re-enumerate: %d", n)
	}
	stop2()
	if n, err := GetDeviceCount(); err != nil || n != 2 {
		t.Errorf("still one reference: %d %v", n, err)
	}
	stop1()
	if _, err := GetDeviceCount(); err == nil {
		t.Error("usable after the last shutdown")
	}
}

// ---- hardware in the loop ------------------------------------------------------

func hw(t testing.TB) func() {
	t.Helper()
	if !amdsmicli.Available() {
		t.Skip("amd-smi not installed")
	}
	if err := Init(); err != nil {
		t.Skipf("no usable GPU: %v", err)
	}
	return func() { _ = Shutdown() }
}

func TestDeviceCount(t *testing.T) {
	defer hw(t)()
	want, err := amdsmicli.DeviceCount()
	if err != nil {
		t.Fatal(err)
	}
	got, err := GetDeviceCount()
	if err != nil || got != want {
		t.Errorf("count: got %d want %d (%v)", got, want, err)
	}
}

func TestDeviceInfo(t *testing.T) {
	defer hw(t)()
	n, _ := GetDeviceCount()
	for i := uint(0); i < n; i++ {
		d, err := NewDevice(i)
		if err != nil {
			t.Fatal(err)
		}
		s, err := amdsmicli.Query("static", int(i))
		if err != nil || len(s) != 1 {
			t.Fatalf("oracle: %v", err)
		}
		o := s[0]
		if !strings.EqualFold(d.PCI.BusID, o["bus.bdf"].(string)) {
			t.Errorf("bus id: %s vs %v", d.PCI.BusID, o["bus.bdf"])
		}
		if d.Memory == nil || int64(*d.Memory) != o["vram.size"].(int64) {
			t.Errorf("memory: %v vs %v", d.Memory, o["vram.size"])
		}
		if int64(d.NumCU) != o["asic.num_compute_units"].(int64) || d.GfxTarget != o["asic.target_graphics_version"].(string) {
			t.Errorf("asic: %d %s", d.NumCU, d.GfxTarget)
		}
		if d.VBIOS != o["ifwi.version"].(string) {
			t.Errorf("vbios: %s vs %v", d.VBIOS, o["ifwi.version"])
		}
	}
}

func within(t *testing.T, what string, a, b *uint, theirs interface{}, tol int64) {
	t.Helper()
	v, ok := theirs.(int64)
	if !ok || a == nil || b == nil {
		return // one side has no value: nothing to compare
	}
	lo, hi := int64(*a), int64(*b)
	if lo > hi {
		lo, hi = hi, lo
	}
	if v < lo-tol || v > hi+tol {
		t.Errorf("%s: oracle %d outside [%d, %d] +- %d", what, v, lo, hi, tol)
	}
}

func TestDeviceStatus(t *testing.T) {
	defer hw(t)()
	d, err := NewDeviceLite(0)
	if err != nil {
		t.Fatal(err)
	}
	before, err := d.Status()
	if err != nil {
		t.Fatal(err)
	}
	m, err := amdsmicli.Query("metric", 0)
	if err != nil {
		t.Fatal(err)
	}
	after, _ := d.Status()
	o := m[0]
	within(t, "temperature", before.Temperature, after.Temperature, o["temperature.hotspot"], 6)
	within(t, "hbm temperature", before.HBMTemp, after.HBMTemp, o["temperature.mem"], 6)
	within(t, "power", before.Power, after.Power, o["power.socket_power"], 60)
	within(t, "gpu utilisation", before.Utilization.GPU, after.Utilization.GPU, o["usage.gfx_activity"], 15)
	within(t, "memory clock", before.Clocks.Memory, after.Clocks.Memory, o["clock.mem_0.clk"], 50)
	if (before.Utilization.Encoder == nil) != (o["usage.mm_activity"] == nil) {
		t.Errorf("encoder: ours %v, oracle %v", before.Utilization.Encoder, o["usage.mm_activity"])
	}
}

func BenchmarkDeviceCount1(b *testing.B) {
	defer hw(b)()
	for i := 0; i < b.N; i++ {
		if _, err := GetDeviceCount(); err != nil {
			b.Fatal(err)
		}
	}
}

func BenchmarkDeviceInfo1(b *testing.B) {
	defer hw(b)()
	for i := 0; i < b.N; i++ {
		if _, err := NewDevice(0); err != nil {
			b.Fatal(err)
		}
	}
}

func BenchmarkDeviceStatus1(b *testing.B) {
	defer hw(b)()
	d, _ := NewDeviceLite(0)
	for i := 0; i < b.N; i++ {
		if _, err := d.Status(); err != nil {
			b.Fatal(err)
		}
	}
}
