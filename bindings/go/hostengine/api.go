package hostengine

import (
	"fmt"
	"sync"
)

var (
	initMu    sync.Mutex
	initCount int
)

// Init starts or connects to the host engine in the given mode.  Calls nest: only
// the first one initialises, and Shutdown must be called once per Init (same
// contract as the reference: bindings/go/dcgm/api.go:19-47; the library keeps the
// same count, so mixing this package with direct C callers stays balanced).
func Init(m mode, args ...string) (cleanup func(), err error) {
	initMu.Lock()
	defer initMu.Unlock()
	if initCount < 0 {
		return nil, fmt.Errorf("hostengine: init count is negative; Shutdown was called without Init")
	}
	if err = initEngine(m, args...); err != nil {
		return nil, err
	}
	initCount++
	return func() { _ = Shutdown() }, nil
}

// Shutdown releases one Init; the engine stops when the last one is released.
func Shutdown() error {
	initMu.Lock()
	defer initMu.Unlock()
	if initCount <= 0 {
		return fmt.Errorf("hostengine: Shutdown without a matching Init")
	}
	initCount--
	if initCount == 0 {
		stopAllPolicyListeners()
	}
	return shutdownEngine()
}

// GetAllDeviceCount counts every GPU the engine sees.
func GetAllDeviceCount() (uint, error) { return getAllDeviceCount() }

// GetSupportedDevices lists the GPUs the engine can fully sample.
func GetSupportedDevices() ([]uint, error) { return getSupportedDevices() }

// GetDeviceInfo describes one GPU: identity, limits, clocks, topology.
func GetDeviceInfo(gpuID uint) (Device, error) { return getDeviceInfo(gpuID) }

// GetDeviceStatus returns the latest cached telemetry of one GPU.
func GetDeviceStatus(gpuID uint) (DeviceStatus, error) { return latestValuesForDevice(gpuID) }

// GetDeviceTopology lists the peers of one GPU and how each is reached.
func GetDeviceTopology(gpuID uint) ([]P2PLink, error) { return getDeviceTopology(gpuID) }

// WatchPidFields starts per-process accounting on all GPUs.
func WatchPidFields() (GroupHandle, error) { return watchPidFields() }

// GetProcessInfo returns what was accounted to pid on each GPU it used.
func GetProcessInfo(group GroupHandle, pid uint) ([]ProcessInfo, error) { return getProcessInfo(group, pid) }

// HealthCheckByGpuId enables every health watch on the GPU and runs a check.
func HealthCheckByGpuId(gpuID uint) (DeviceHealth, error) { return healthCheckByGpuID(gpuID) }

// Policy registers the given conditions on one GPU and returns the channel
// violations arrive on.  No conditions means all of them.
func Policy(gpuID uint, conditions ...policyCondition) (<-chan PolicyViolation, error) {
	return registerPolicy(gpuID, DefaultThresholds(), conditions...)
}

// PolicyWithThresholds is Policy with explicit thresholds.
func PolicyWithThresholds(gpuID uint, t Thresholds, conditions ...policyCondition) (<-chan PolicyViolation, error) {
	return registerPolicy(gpuID, t, conditions...)
}

// Introspect reports the engine's own memory and CPU use.
func Introspect() (Status, error) { return introspect() }
