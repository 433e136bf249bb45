package hostengine

/*
#include "capi.h"
*/
import "C"

// XgmiLink is one physical xGMI link of a GPU: its state, the GPU at the far end
// and its traffic counters — what the reference collects by walking the NVLink
// indices of a device (bindings/go/nvml/bindings.go:268-318).  Optional values are
// nil where the platform does not report them.
type XgmiLink struct {
	Link       uint
	Up         *bool   // nil: not wired or not reported
	PeerGPU    *uint   // engine index of the GPU at the far end
	PeerBusID  string  // empty when unknown
	BitRateGbs *int64  // per lane
	MaxBwGbs   *int64  // advertised, one direction
	RxKB, TxKB *uint64 // cumulative
}

func optBool(state C.int32_t) *bool {
	if state < 0 {
		return nil
	}
	b := state != 0
	return &b
}

func optU64(v C.int64_t) *uint64 {
	if v < 0 {
		return nil
	}
	x := uint64(v)
	return &x
}

// GetXgmiLinks lists every link of the GPU, in link-index order.
func GetXgmiLinks(gpuID uint) ([]XgmiLink, error) {
	var raw [C.GM_MAX_XGMI_LINKS]C.gm_xgmi_link_t
	n := C.int(C.GM_MAX_XGMI_LINKS)
	if err := check("xgmi links", C.gm_xgmi_links(C.uint32_t(gpuID), &raw[0], &n)); err != nil {
		return nil, err
	}
	out := make([]XgmiLink, 0, int(n))
	for i := 0; i < int(n); i++ {
		l := &raw[i]
		out = append(out, XgmiLink{Link: uint(l.link), Up: optBool(l.state), PeerGPU: optUint(C.int64_t(l.peer_gpu), l.peer_gpu >= 0),
			PeerBusID: C.GoString(&l.peer_bdf[0]), BitRateGbs: optInt(l.bit_rate_gbps, l.bit_rate_gbps >= 0),
			MaxBwGbs: optInt(l.max_bandwidth_gbps, l.max_bandwidth_gbps >= 0), RxKB: optU64(l.rx_kb), TxKB: optU64(l.tx_kb)})
	}
	return out, nil
}
