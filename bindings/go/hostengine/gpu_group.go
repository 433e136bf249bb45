package hostengine

/*
#include "capi.h"
*/
import "C"

// GroupHandle names a set of GPUs inside the engine.
type GroupHandle struct{ handle C.gm_group_t }

// GroupAllGPUs is the built-in group of every GPU.
var GroupAllGPUs = GroupHandle{handle: C.GM_GROUP_ALL_GPUS}

// CreateGroup makes an empty named group; names are unique per engine.
func CreateGroup(name string) (GroupHandle, error) {
	c, free := cstr(name)
	defer free()
	var g C.gm_group_t
	err := check("group create", C.gm_group_create(c, &g))
	return GroupHandle{g}, err
}

// NewDefaultGroup makes a group holding every GPU.
func NewDefaultGroup(name string) (GroupHandle, error) {
	g, err := CreateGroup(name)
	if err != nil {
		return g, err
	}
	n, err := getAllDeviceCount()
	if err != nil {
		return g, err
	}
	for i := uint(0); i < n; i++ {
		if err = AddToGroup(g, i); err != nil {
			_ = DestroyGroup(g)
			return g, err
		}
	}
	return g, nil
}

func AddToGroup(g GroupHandle, gpuID uint) error {
	return check("group add", C.gm_group_add_device(g.handle, C.uint32_t(gpuID)))
}

func DestroyGroup(g GroupHandle) error { return check("group destroy", C.gm_group_destroy(g.handle)) }
