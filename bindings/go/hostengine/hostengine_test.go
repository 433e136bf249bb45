package hostengine

// Fake-backend tests run anywhere; the hardware tests compare the embedded engine
// with `amd-smi` the way the reference compares DCGM with nvidia-smi
// (bindings/go/dcgm/dcgm_test.go:18-190) and are skipped without a GPU.

import (
	"math"
	"strings"
	"testing"
	"time"

	"github.com/k8s-gpu-monitor-amd/gpumon/bindings/go/smi/amdsmicli"
)

func fakeCtl(t testing.TB, command string) {
	t.Helper()
	if err := FakeCtl(command); err != nil {
		t.Fatalf("fake ctl %q: %v", command, err)
	}
}

func withFake(t testing.TB, spec string) func() {
	t.Helper()
	cleanup, err := Init(Embedded, "fake", spec)
	if err != nil {
		t.Fatalf("init: %v", err)
	}
	return cleanup
}

func TestInitShutdownAreCounted(t *testing.T) {
	if err := Shutdown(); err == nil {
		t.Fatal("Shutdown without Init must fail")
	}
	c1 := withFake(t, "gpus=3")
	c2 := withFake(t, "gpus=8")
	if n, _ := GetAllDeviceCount(); n != 3 {
		t.Errorf("nested Init must not restart the engine: %d GPUs", n)
	}
	c2()
	if n, err := GetAllDeviceCount(); err != nil || n != 3 {
		t.Errorf("one reference left: %d, %v", n, err)
	}
	c1()
	if _, err := GetAllDeviceCount(); err == nil {
		t.Error("engine still usable after the last Shutdown")
	}
}

func TestDeviceInfoAndStatusOnFake(t *testing.T) {
	defer withFake(t, "gpus=4,seed=2,steady=1")()
	ids, err := GetSupportedDevices()
	if err != nil || len(ids) != 4 {
		t.Fatalf("supported: %v %v", ids, err)
	}
	d, err := GetDeviceInfo(1)
	if err != nil {
		t.Fatal(err)
	}
	if d.Identifiers.Model != "AMD Instinct MI355X" || d.PCI.VRAMTotal == nil || *d.PCI.VRAMTotal != 294896 {
		t.Errorf("device: %+v", d)
	}
	if len(d.Topology) != 3 || d.Topology[0].Name != "XGMI1" || d.Topology[0].BusID == d.PCI.BusID {
		t.Errorf("topology must list the PEERS' bus ids: %+v", d.Topology)
	}
	st, err := GetDeviceStatus(1)
	if err != nil {
		t.Fatal(err)
	}
	if st.Temperature == nil || st.Power == nil || st.Utilization.Encoder != nil || st.FanSpeed != nil {
		t.Errorf("status: %+v", st)
	}
	if _, err := GetDeviceStatus(9); err == nil {
		t.Error("bad index accepted")
	}
}

func TestWatchesAndHistory(t *testing.T) {
	defer withFake(t, "gpus=2,seed=2")()
	g, err := NewDefaultGroup("t")
	if err != nil {
		t.Fatal(err)
	}
	f, err := FieldGroupCreate("t", []uint16{150, 155, 50})
	if err != nil {
		t.Fatal(err)
	}
	if err = WatchFieldsWithOptions(g, f, 20*time.Millisecond, time.Second, 0); err != nil {
		t.Fatal(err)
	}
	_ = UpdateAllFields()
	time.Sleep(200 * time.Millisecond)
	v, err := GetLatestValues(0, []uint16{150, 155, 50, 206})
	if err != nil || len(v) != 4 {
		t.Fatal(err)
	}
	if _, ok := v[0].V.(int64); !ok {
		t.Errorf("temperature is an integer field: %T", v[0].V)
	}
	if _, ok := v[1].V.(float64); !ok {
		t.Errorf("power is a float field: %T", v[1].V)
	}
	if v[2].V != "AMD Instinct MI355X" || v[3].V != nil {
		t.Errorf("name / blank encoder: %v %v", v[2].V, v[3].V)
	}
	h, err := GetValuesSince(0, 150, time.Time{})
	if err != nil || len(h) < 5 {
		t.Errorf("history: %d samples, %v", len(h), err)
	}
	if err = UnwatchFields(g, f); err != nil {
		t.Error(err)
	}
	_ = FieldGroupDestroy(f)
	_ = DestroyGroup(g)
}

func TestHealthOnFake(t *testing.T) {
	defer withFake(t, "gpus=2")()
	h, err := HealthCheckByGpuId(0)
	if err != nil || h.Status != "Healthy" || len(h.Watches) != 0 {
		t.Fatalf("healthy GPU: %+v %v", h, err)
	}
	fakeCtl(t, "link_down gpu=0 link=1")
	_ = UpdateAllFields()
	h, _ = HealthCheckByGpuId(0)
	if h.Status != "Failure" || len(h.Watches) == 0 || h.Watches[0].Type != "xGMI" {
		t.Errorf("link down: %+v", h)
	}
}

func TestPolicyViolationsArriveOnTheChannel(t *testing.T) {
	defer withFake(t, "gpus=2")()
	ch, err := PolicyWithThresholds(1, Thresholds{MaxBadPages: 10, ThermalC: 90, PowerW: 1400, HysteresisPct: 3}, ThermalPolicy, GpuEventPolicy)
	if err != nil {
		t.Fatal(err)
	}
	fakeCtl(t, "set gpu=1 field=150 value=97")
	_ = UpdateAllFields()
	select {
	case v := <-ch:
		d, ok := v.Data.(ThermalData)
		if v.Condition != ThermalPolicy || v.GPU != 1 || !ok || d.ThermalViolation != 97 {
			t.Errorf("violation: %+v", v)
		}
	case <-time.After(3 * time.Second):
		t.Fatal("no thermal violation")
	}
	_ = UpdateAllFields()
	select {
	case v := <-ch:
		t.Errorf("exactly once per crossing, got a second one: %+v", v)
	case <-time.After(200 * time.Millisecond):
	}
	fakeCtl(t, "event gpu=1 code=4 msg=reset")
	select {
	case v := <-ch:
		if d, ok := v.Data.(GpuEventData); !ok || d.Code != 4 || v.Condition != GpuEventPolicy {
			t.Errorf("event violation: %+v", v)
		}
	case <-time.After(3 * time.Second):
		t.Fatal("no event violation")
	}
	if _, err := Policy(1, policyCondition("nonsense")); err == nil {
		t.Error("unknown condition accepted")
	}
}

func TestProcessInfoAndIntrospection(t *testing.T) {
	defer withFake(t, "gpus=2,load=0.8")()
	g, err := WatchPidFields()
	if err != nil {
		t.Fatal(err)
	}
	fakeCtl(t, "proc_add gpu=1 pid=555 name=train")
	for i := 0; i < 5; i++ {
		_ = UpdateAllFields()
		time.Sleep(20 * time.Millisecond)
	}
	p, err := GetProcessInfo(g, 555)
	if err != nil || len(p) != 1 || p[0].GPU != 1 || p[0].Name != "train" {
		t.Fatalf("process info: %+v %v", p, err)
	}
	if p[0].ProcessUtilization.EndTime.String() != "Running" || p[0].Violations.SyncBoost != nil {
		t.Errorf("running process / AMD has no sync boost: %+v", p[0])
	}
	if _, err := GetProcessInfo(g, 99999); err == nil {
		t.Error("unknown pid must be an error")
	}
	st, err := Introspect()
	if err != nil || st.Memory < 1000 || st.SampleCycles == 0 || math.IsNaN(st.CPU) {
		t.Errorf("introspection: %+v %v", st, err)
	}
}

func TestLinksThresholdsAndDiagResultsOnFake(t *testing.T) {
	defer withFake(t, "gpus=8,seed=1,steady=1")()
	links, err := GetXgmiLinks(2)
	if err != nil || len(links) != 7 {
		t.Fatalf("links: %v %v", links, err)
	}
	peers := map[uint]bool{}
	for _, l := range links {
		if l.Up == nil || !*l.Up || l.PeerGPU == nil || *l.PeerGPU == 2 || l.PeerBusID == "" {
			t.Errorf("link %d: %+v", l.Link, l)
			continue
		}
		peers[*l.PeerGPU] = true
	}
	if len(peers) != 7 {
		t.Errorf("a full mesh has seven distinct peers: %v", peers)
	}

	th, err := GetHealthThresholds()
	if err != nil || th.HotspotFailC != 110 || th.MemWarnC != 95 {
		t.Fatalf("default thresholds: %+v %v", th, err)
	}
	th.HotspotWarnC = 60
	if err := SetHealthThresholds(th); err != nil {
		t.Fatal(err)
	}
	if got, _ := GetHealthThresholds(); got.HotspotWarnC != 60 {
		t.Errorf("thresholds not kept: %+v", got)
	}
	th.HotspotWarnC, th.HotspotFailC = 120, 110
	if err := SetHealthThresholds(th); err == nil {
		t.Error("warn above fail must be refused")
	}

	if err := SubmitDiagResult(1, "xgmi_p2p", HealthXGMI, DiagFail, 41.0, "link 3 to GPU 4: 41.0 GB/s"); err != nil {
		t.Fatal(err)
	}
	res, err := GetDiagResults(1)
	if err != nil || len(res) != 1 || res[0].Health != "Failure" || res[0].Value == nil || math.Abs(*res[0].Value-41.0) > 1e-9 {
		t.Fatalf("diag results: %+v %v", res, err)
	}
	h, err := HealthCheckByGpuId(1)
	if err != nil || h.Status != "Failure" {
		t.Fatalf("a failed diagnostic must fail the health check: %+v %v", h, err)
	}
	found := false
	for _, w := range h.Watches {
		found = found || strings.Contains(w.Error, "link 3 to GPU 4")
	}
	if !found {
		t.Errorf("incident text missing: %+v", h.Watches)
	}
}

func TestStartHostengineMode(t *testing.T) {
	cleanup, err := Init(StartHostengine, "", "fake", "gpus=2")
	if err != nil {
		t.Skipf("gpumon-hostengine not found: %v", err)
	}
	defer cleanup()
	if n, err := GetAllDeviceCount(); err != nil || n != 2 {
		t.Errorf("child engine: %d %v", n, err)
	}
}

// ---- hardware in the loop ------------------------------------------------------

func hw(t testing.TB) func() {
	t.Helper()
	if !amdsmicli.Available() {
		t.Skip("amd-smi not installed")
	}
	cleanup, err := Init(Embedded)
	if err != nil {
		t.Skipf("no usable GPU: %v", err)
	}
	return cleanup
}

func TestDeviceCount(t *testing.T) {
	defer hw(t)()
	want, err := amdsmicli.DeviceCount()
	if err != nil {
		t.Fatal(err)
	}
	if got, err := GetAllDeviceCount(); err != nil || got != want {
		t.Errorf("count: got %d want %d (%v)", got, want, err)
	}
}

func TestDeviceInfo(t *testing.T) {
	defer hw(t)()
	n, _ := GetAllDeviceCount()
	for i := uint(0); i < n; i++ {
		d, err := GetDeviceInfo(i)
		if err != nil {
			t.Fatal(err)
		}
		s, err := amdsmicli.Query("static", int(i))
		if err != nil || len(s) != 1 {
			t.Fatal(err)
		}
		o := s[0]
		if !strings.EqualFold(d.PCI.BusID, o["bus.bdf"].(string)) || d.Identifiers.Vbios != o["ifwi.version"].(string) {
			t.Errorf("identity: %+v", d.Identifiers)
		}
		if d.PCI.VRAMTotal == nil || int64(*d.PCI.VRAMTotal) != o["vram.size"].(int64) {
			t.Errorf("HBM size: %v vs %v", d.PCI.VRAMTotal, o["vram.size"])
		}
		if !strings.HasPrefix(o["driver.version"].(string), d.Identifiers.DriverVersion) {
			t.Errorf("driver: %q", d.Identifiers.DriverVersion)
		}
	}
}

func TestDeviceStatus(t *testing.T) {
	defer hw(t)()
	before, err := GetDeviceStatus(0)
	if err != nil {
		t.Fatal(err)
	}
	m, err := amdsmicli.Query("metric", 0)
	if err != nil {
		t.Fatal(err)
	}
	time.Sleep(1100 * time.Millisecond) // one engine period
	after, _ := GetDeviceStatus(0)
	o := m[0]
	if hot, ok := o["temperature.hotspot"].(int64); ok && before.Temperature != nil && after.Temperature != nil {
		lo, hi := int64(*before.Temperature), int64(*after.Temperature)
		if lo > hi {
			lo, hi = hi, lo
		}
		if hot < lo-6 || hot > hi+6 {
			t.Errorf("temperature: oracle %d, ours %d..%d", hot, lo, hi)
		}
	}
	if w, ok := o["power.socket_power"].(int64); ok && before.Power != nil && after.Power != nil {
		lo, hi := math.Min(*before.Power, *after.Power), math.Max(*before.Power, *after.Power)
		if float64(w) < lo-60 || float64(w) > hi+60 {
			t.Errorf("power: oracle %d, ours %.0f..%.0f", w, lo, hi)
		}
	}
}

func BenchmarkDeviceCount1(b *testing.B) {
	defer hw(b)()
	for i := 0; i < b.N; i++ {
		if _, err := GetAllDeviceCount(); err != nil {
			b.Fatal(err)
		}
	}
}

func BenchmarkDeviceInfo1(b *testing.B) {
	defer hw(b)()
	for i := 0; i < b.N; i++ {
		if _, err := GetDeviceInfo(0); err != nil {
			b.Fatal(err)
		}
	}
}

func BenchmarkDeviceStatus1(b *testing.B) {
	defer hw(b)()
	for i := 0; i < b.N; i++ {
		if _, err := GetDeviceStatus(0); err != nil {
			b.Fatal(err)
		}
	}
}
