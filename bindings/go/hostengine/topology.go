package hostengine

/*
#include "capi.h"
*/
import "C"

// P2PLinkType classifies the path between two GPUs: the PCIe classes of the
// reference (bindings/go/dcgm/topology.go:15-56) plus xGMI with a hop count
// where the reference has NV1..NV4.
type P2PLinkType uint

const (
	P2PLinkUnknown P2PLinkType = iota
	P2PLinkCrossCPU
	P2PLinkSameCPU
	P2PLinkHostBridge
	P2PLinkMultiSwitch
	P2PLinkSingleSwitch
	P2PLinkSameBoard
	P2PLinkXGMI
)

// PCIPaths is the short code of the link class as a topology matrix prints it: PSB, PIX, PXB, PHB, NODE, SYS as in
// the reference (bindings/go/dcgm/topology.go:31-56), XGMI where the reference has NV1..NV4 (P2PLink.Name adds the
// hop count), N/A when unknown.
func (l P2PLinkType) PCIPaths() string { return C.GoString(C.gm_link_type_name(C.int32_t(l), 0)) }

// Description is the long form ("Connected through the xGMI fabric", "Single PCI switch", ...).
func (l P2PLinkType) Description() string { return C.GoString(C.gm_link_type_description(C.int32_t(l))) }

func (l P2PLinkType) String() string { return l.PCIPaths() }

type P2PLink struct {
	GPU      uint
	BusID    string // the PEER's bus id
	Link     P2PLinkType
	Hops     uint
	Name     string // "XGMI1", "PIX", "SYS", ...
	MaxMBs   *int64 // theoretical link bandwidth when the platform reports it
	Directly bool   // peer-to-peer accessible
}

func getDeviceTopology(gpuID uint) ([]P2PLink, error) {
	var t C.gm_topology_t
	C.he_topology(&t)
	if err := check("device topology", C.gm_device_topology(C.uint32_t(gpuID), &t)); err != nil {
		return nil, err
	}
	out := make([]P2PLink, 0, int(t.num_peers))
	for i := 0; i < int(t.num_peers); i++ {
		p := &t.peers[i]
		out = append(out, P2PLink{GPU: uint(p.gpu), BusID: C.GoString(&p.pci_bdf[0]), Link: P2PLinkType(p.link),
			Hops: uint(p.hops), Name: C.GoString(C.gm_link_type_name(p.link, p.hops)),
			MaxMBs: optInt(p.max_bw_mbs, p.max_bw_mbs >= 0), Directly: p.p2p_accessible != 0})
	}
	return out, nil
}
