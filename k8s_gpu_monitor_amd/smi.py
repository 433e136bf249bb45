"""Direct device layer: every call is one synchronous round to the device
library — no host engine, no sampler thread, no cache.

The counterpart of the reference's ``nvml`` Go package (bindings/go/nvml/nvml.go:
``Init``/``Shutdown`` :274-280, ``GetDeviceCount``, ``GetDriverVersion``,
``NewDevice`` :328-396, ``NewDeviceLite`` :398-431, ``(*Device).Status`` :433-512,
``GetP2PLink`` :514-537, ``GetDeviceMode`` :582-604, event sets bindings.go:68-146),
kept separate from :mod:`engine` the way that package is separate from ``dcgm``.
The function names match :mod:`engine`'s, so :mod:`models.device` can be built on
either (``new_device(i, api=smi)``).
"""
import contextlib
import ctypes as C

from . import _abi as A
from ._lib import GpumonError, check, lib

_typed = False


def _L():
    global _typed
    L = lib()
    if not _typed:
        st, u32p = C.c_int, C.POINTER(C.c_uint32)
        sig = {
            "gm_smi_init": [C.c_char_p, C.c_char_p], "gm_smi_shutdown": [], "gm_smi_device_count": [u32p],
            "gm_smi_driver_version": [C.c_char_p, C.c_size_t],
            "gm_smi_runtime_version": [u32p, u32p, C.c_char_p, C.c_size_t],
            "gm_smi_device_attributes": [C.c_uint32, C.POINTER(A.DeviceAttributes)],
            "gm_smi_device_status": [C.c_uint32, C.POINTER(A.DeviceStatus)],
            "gm_smi_read_fields": [C.c_uint32, C.POINTER(C.c_uint16), C.c_int, C.POINTER(A.Value)],
            "gm_smi_device_processes": [C.c_uint32, C.POINTER(A.Process), C.POINTER(C.c_int)],
            "gm_smi_peer_link": [C.c_uint32, C.c_uint32, C.POINTER(A.PeerLink)],
            "gm_smi_xgmi_links": [C.c_uint32, C.POINTER(A.XgmiLink), C.POINTER(C.c_int)],
            "gm_smi_device_mode": [C.c_uint32, C.POINTER(A.DeviceMode)],
            "gm_smi_event_wait": [C.c_int32, C.POINTER(A.Event)], "gm_smi_fake_ctl": [C.c_char_p],
        }
        for name, args in sig.items():
            fn = getattr(L, name)
            fn.restype, fn.argtypes = st, args
        _typed = True
    return L


def _b(s):
    return None if s is None else s.encode()


def init(backend=None, backend_spec=None):
    check(_L().gm_smi_init(_b(backend), _b(backend_spec)), "smi.init")


def shutdown():
    check(_L().gm_smi_shutdown(), "smi.shutdown")


@contextlib.contextmanager
def session(**kw):
    init(**kw)
    try:
        yield
    finally:
        shutdown()


def device_count():
    n = C.c_uint32()
    check(_L().gm_smi_device_count(C.byref(n)), "smi.device_count")
    return n.value


def driver_version():
    buf = C.create_string_buffer(A.GM_STR_LEN)
    check(_L().gm_smi_driver_version(buf, len(buf)), "smi.driver_version")
    return buf.value.decode()


def runtime_version():
    """(major, minor, text) of the ROCm release under the device library, or None when unknown: the
    counterpart of nvml.GetCudaDriverVersion (bindings/go/nvml/nvml.go:290-292)."""
    major, minor = C.c_uint32(), C.c_uint32()
    buf = C.create_string_buffer(A.GM_STR_LEN)
    st = _L().gm_smi_runtime_version(C.byref(major), C.byref(minor), buf, len(buf))
    if st == A.GM_ST_NOT_SUPPORTED:
        return None
    check(st, "smi.runtime_version")
    return major.value, minor.value, buf.value.decode()


def _struct_dict(s, skip=("version",)):
    out = {}
    for name, _ in s._fields_:
        if name in skip:
            continue
        v = getattr(s, name)
        out[name] = v.decode() if isinstance(v, bytes) else v
    return out


def device_attributes(gpu):
    a = A.versioned(A.DeviceAttributes)
    check(_L().gm_smi_device_attributes(gpu, C.byref(a)), f"smi.device_attributes({gpu})")
    return _struct_dict(a)


def device_status(gpu):
    d = A.versioned(A.DeviceStatus)
    check(_L().gm_smi_device_status(gpu, C.byref(d)), f"smi.device_status({gpu})")
    return d.as_dict()


def read_fields(gpu, field_ids):
    """{field_id: value or None} straight from the device library."""
    arr = (C.c_uint16 * len(field_ids))(*field_ids)
    out = (A.Value * len(field_ids))()
    check(_L().gm_smi_read_fields(gpu, arr, len(field_ids), out), "smi.read_fields")
    return {fid: out[i].py() for i, fid in enumerate(field_ids)}


def device_processes(gpu):
    buf = (A.Process * A.GM_MAX_PROCS)()
    n = C.c_int(A.GM_MAX_PROCS)
    st = _L().gm_smi_device_processes(gpu, buf, C.byref(n))
    if st not in (A.GM_ST_OK, A.GM_ST_INSUFFICIENT_SIZE):
        raise GpumonError(st, "smi.device_processes")
    return [_struct_dict(buf[i], skip=()) for i in range(min(n.value, A.GM_MAX_PROCS))]


def peer_link(gpu, peer):
    p = A.PeerLink()
    check(_L().gm_smi_peer_link(gpu, peer, C.byref(p)), f"smi.peer_link({gpu},{peer})")
    d = _struct_dict(p, skip=())
    d["link_name"] = lib().gm_link_type_name(p.link, p.hops).decode()
    return d


def xgmi_links(gpu):
    """Every physical xGMI link of the GPU, straight from the device library (same shape as engine.xgmi_links)."""
    buf = (A.XgmiLink * 64)()
    n = C.c_int(64)
    check(_L().gm_smi_xgmi_links(gpu, buf, C.byref(n)), f"smi.xgmi_links({gpu})")
    out = []
    for i in range(n.value):
        d = _struct_dict(buf[i], skip=())
        for k in ("bit_rate_gbps", "max_bandwidth_gbps", "rx_kb", "tx_kb"):
            if d[k] < 0:
                d[k] = None
        d["peer_gpu"] = None if d["peer_gpu"] < 0 else d["peer_gpu"]
        out.append(d)
    return out


def device_topology(gpu):
    """Same shape as engine.device_topology; an unsupported platform yields an empty peer list."""
    a = device_attributes(gpu)
    peers = []
    for other in range(device_count()):
        if other == gpu:
            continue
        try:
            peers.append(peer_link(gpu, other))
        except GpumonError as e:
            if e.status == A.GM_ST_NOT_SUPPORTED:
                return {"gpu": gpu, "cpu_affinity": a["cpu_affinity"], "numa_node": a["numa_node"], "peers": []}
            raise
    return {"gpu": gpu, "cpu_affinity": a["cpu_affinity"], "numa_node": a["numa_node"], "peers": peers}


def device_mode(gpu):
    m = A.versioned(A.DeviceMode)
    check(_L().gm_smi_device_mode(gpu, C.byref(m)), "smi.device_mode")
    return _struct_dict(m)


def event_wait(timeout_ms=1000):
    """Next driver event as a dict, or None on timeout (reference: WaitForEvent, bindings.go:120-141)."""
    e = A.Event()
    st = _L().gm_smi_event_wait(timeout_ms, C.byref(e))
    if st == A.GM_ST_TIMEOUT:
        return None
    check(st, "smi.event_wait")
    d = _struct_dict(e, skip=())
    d["name"] = lib().gm_event_code_name(e.code).decode()
    return d


def fake_ctl(command):
    check(_L().gm_smi_fake_ctl(_b(command)), f"smi.fake_ctl({command})")
