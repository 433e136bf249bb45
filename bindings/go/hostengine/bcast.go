package hostengine

import "sync"

// fanout delivers every published violation to all current subscribers.  Each
// subscriber has a buffered channel; a subscriber that stops reading loses its
// oldest pending violations rather than blocking the engine's notifier thread or
// the other subscribers.  (The reference's broadcaster starts one goroutine per
// send and can block forever on a stalled reader: bindings/go/dcgm/bcast.go:67-80.)
type fanout struct {
	mu   sync.Mutex
	subs map[chan PolicyViolation]struct{}
}

const subscriberDepth = 64

func newFanout() *fanout { return &fanout{subs: map[chan PolicyViolation]struct{}{}} }

func (f *fanout) subscribe() chan PolicyViolation {
	ch := make(chan PolicyViolation, subscriberDepth)
	f.mu.Lock()
	f.subs[ch] = struct{}{}
	f.mu.Unlock()
	return ch
}

func (f *fanout) unsubscribe(ch chan PolicyViolation) {
	f.mu.Lock()
	if _, ok := f.subs[ch]; ok {
		delete(f.subs, ch)
		close(ch)
	}
	f.mu.Unlock()
}

func (f *fanout) publish(v PolicyViolation) {
	f.mu.Lock()
	defer f.mu.Unlock()
	for ch := range f.subs {
		select {
		case ch <- v:
		default: // full: drop the oldest, keep the newest
			select {
			case <-ch:
			default:
			}
			select {
			case ch <- v:
			default:
			}
		}
	}
}

func (f *fanout) closeAll() {
	f.mu.Lock()
	for ch := range f.subs {
		delete(f.subs, ch)
		close(ch)
	}
	f.mu.Unlock()
}
