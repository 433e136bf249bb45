// Event sets: the shape of the reference's NVML event API (NewEventSet, RegisterEvent,
// RegisterEventForDevice, WaitForEvent, DeleteEventSet: bindings/go/nvml/bindings.go:68-146)
// over the direct layer's one driver-event channel.
//
// The device library delivers every event of every GPU on one channel, so a set is a filter
// kept on this side: a mask of event types for all devices plus a mask per registered
// device.  Event types are bits, 1 << GM_EVENT_* code; XidCriticalError stands for every
// critical driver event (VM fault, reset, uncorrectable RAS error, poison, ring hang), the
// counterpart of the one type the reference registers for.
package smi

import (
	"fmt"
	"sync"
	"time"
)

const (
	EventVMFault         = 1 << 1
	EventThermalThrottle = 1 << 2
	EventGPUPreReset     = 1 << 3
	EventGPUPostReset    = 1 << 4
	EventRASUncorrectable = 1 << 5
	EventRASPoison       = 1 << 6
	EventRingHang        = 1 << 7
	EventProcessStart    = 1 << 20
	EventProcessEnd      = 1 << 21

	// XidCriticalError is what the reference's samples wait for (nvml.h:1103).
	XidCriticalError = EventVMFault | EventGPUPreReset | EventGPUPostReset | EventRASUncorrectable | EventRASPoison | EventRingHang
)

type eventSet struct {
	mu        sync.Mutex
	all       uint64
	perDevice map[string]uint64
	deleted   bool
}

// EventSet is a handle, as in the reference; the zero value is not usable.
type EventSet struct{ set *eventSet }

func NewEventSet() EventSet {
	return EventSet{set: &eventSet{perDevice: map[string]uint64{}}}
}

func (es EventSet) live(op string) (*eventSet, error) {
	if es.set == nil {
		return nil, fmt.Errorf("smi: %s: event set was not created with NewEventSet", op)
	}
	return es.set, nil
}

// RegisterEvent adds the event types of the mask for every GPU.
func RegisterEvent(es EventSet, event int) error {
	s, err := es.live("RegisterEvent")
	if err != nil {
		return err
	}
	if event <= 0 {
		return fmt.Errorf("smi: RegisterEvent: empty event mask")
	}
	s.mu.Lock()
	defer s.mu.Unlock()
	if s.deleted {
		return fmt.Errorf("smi: RegisterEvent: event set was deleted")
	}
	s.all |= uint64(event)
	return nil
}

// RegisterEventForDevice adds the event types of the mask for the GPU with this UUID; an
// unknown UUID is an error, as in the reference (bindings.go:97-129).
func RegisterEventForDevice(es EventSet, event int, uuid string) error {
	s, err := es.live("RegisterEventForDevice")
	if err != nil {
		return err
	}
	if event <= 0 {
		return fmt.Errorf("smi: RegisterEventForDevice: empty event mask")
	}
	n, err := deviceCount()
	if err != nil {
		return err
	}
	found := false
	for i := uint(0); i < n && !found; i++ {
		a, err := deviceAttributes(i)
		if err != nil {
			return err
		}
		found = a.uuid == uuid
	}
	if !found {
		return fmt.Errorf("smi: RegisterEventForDevice: device not found with UUID %s", uuid)
	}
	s.mu.Lock()
	defer s.mu.Unlock()
	if s.deleted {
		return fmt.Errorf("smi: RegisterEventForDevice: event set was deleted")
	}
	s.perDevice[uuid] |= uint64(event)
	return nil
}

// DeleteEventSet ends the set; a WaitForEvent on it returns with an error.
func DeleteEventSet(es EventSet) {
	if es.set == nil {
		return
	}
	es.set.mu.Lock()
	es.set.deleted = true
	es.set.mu.Unlock()
}

func (s *eventSet) wants(e *rawEvent) (bool, error) {
	s.mu.Lock()
	defer s.mu.Unlock()
	if s.deleted {
		return false, fmt.Errorf("smi: WaitForEvent: event set was deleted")
	}
	bit := uint64(1) << (e.code & 63)
	return s.all&bit != 0 || s.perDevice[e.uuid]&bit != 0, nil
}

// WaitForEvent blocks up to timeout milliseconds for an event the set is registered for.
// On timeout it returns an Event with Etype 0 and an *Error whose Status is StatusTimeout
// (the reference returns NVML_ERROR_TIMEOUT the same way, bindings.go:134-146).
func WaitForEvent(es EventSet, timeout uint) (Event, error) {
	s, err := es.live("WaitForEvent")
	if err != nil {
		return Event{}, err
	}
	deadline := time.Now().Add(time.Duration(timeout) * time.Millisecond)
	for {
		left := time.Until(deadline)
		if left < 0 {
			left = 0
		}
		// Short slices, so that DeleteEventSet from another goroutine ends the wait.
		slice := left
		if slice > 200*time.Millisecond {
			slice = 200 * time.Millisecond
		}
		e, err := eventWait(int(slice / time.Millisecond))
		if err != nil && !isStatus(err, StatusTimeout) {
			return Event{}, err
		}
		if err == nil {
			ok, werr := s.wants(e)
			if werr != nil {
				return Event{}, werr
			}
			if ok {
				return makeEvent(e), nil
			}
		} else if _, werr := s.wants(&rawEvent{}); werr != nil {
			return Event{}, werr
		}
		if time.Now().After(deadline) || left == 0 {
			return Event{}, &Error{Op: "event wait", Status: StatusTimeout, Msg: "timeout"}
		}
	}
}

func makeEvent(e *rawEvent) Event {
	return Event{GPU: e.gpu, UUID: e.uuid, Code: e.code, Name: eventName(e.code), Message: e.msg,
		Time: time.UnixMicro(e.tsMicros), Etype: uint64(1) << (e.code & 63), Edata: uint64(e.code)}
}
