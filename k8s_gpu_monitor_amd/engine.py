"""Pythonic façade over the C ABI: the same API shape as the reference's Go
``dcgm`` package (bindings/go/dcgm/api.go:19-98) — Init/Shutdown with three
modes, device info / status / topology, health, policies, process accounting,
introspection — plus the exporter controls.
"""
import ctypes as C
from contextlib import contextmanager

from . import _abi as A
from ._lib import GpumonError, check, lib

EMBEDDED, STANDALONE, START_HOSTENGINE = A.GM_MODE_EMBEDDED, A.GM_MODE_STANDALONE, A.GM_MODE_START_HOSTENGINE


def _b(s):
    return s.encode() if isinstance(s, str) else s


def init(mode=EMBEDDED, address=None, is_unix_socket=False, backend=None, backend_spec=None, hostengine_path=None):
    p = A.versioned(A.InitParams)
    p.mode = mode
    p.address = _b(address)
    p.address_is_unix_socket = 1 if is_unix_socket else 0
    p.backend = _b(backend)
    p.backend_spec = _b(backend_spec)
    p.hostengine_path = _b(hostengine_path)
    check(lib().gm_init(C.byref(p)), "gm_init")


def shutdown():
    check(lib().gm_shutdown(), "gm_shutdown")


@contextmanager
def session(**kw):
    init(**kw)
    try:
        yield
    finally:
        shutdown()


def device_count():
    n = C.c_uint32()
    check(lib().gm_device_count(C.byref(n)))
    return n.value


def supported_devices():
    ids = (C.c_uint32 * A.GM_MAX_GPUS)()
    n = C.c_int(A.GM_MAX_GPUS)
    check(lib().gm_supported_devices(ids, C.byref(n)))
    return list(ids[: n.value])


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
    check(lib().gm_device_attributes(gpu, C.byref(a)), f"device_attributes({gpu})")
    return _struct_dict(a)


def device_topology(gpu):
    t = A.versioned(A.Topology)
    check(lib().gm_device_topology(gpu, C.byref(t)), f"device_topology({gpu})")
    peers = []
    for i in range(t.num_peers):
        p = _struct_dict(t.peers[i], skip=())
        p["link_name"] = lib().gm_link_type_name(t.peers[i].link, t.peers[i].hops).decode()
        peers.append(p)
    return {"gpu": t.gpu, "cpu_affinity": t.cpu_affinity.decode(), "numa_node": t.numa_node, "peers": peers}


def xgmi_links(gpu):
    """Every physical xGMI link of the GPU: state (1 up, 0 down, -1 unknown), the GPU and PCI address at the far
    end, rate and traffic counters (None where the platform does not say)."""
    buf = (A.XgmiLink * 64)()
    n = C.c_int(64)
    check(lib().gm_xgmi_links(gpu, buf, C.byref(n)), f"xgmi_links({gpu})")
    out = []
    for i in range(n.value):
        d = _struct_dict(buf[i], skip=())
        for k in ("bit_rate_gbps", "max_bandwidth_gbps", "rx_kb", "tx_kb"):
            if d[k] < 0:
                d[k] = None
        d["peer_gpu"] = None if d["peer_gpu"] < 0 else d["peer_gpu"]
        out.append(d)
    return out


def device_status(gpu):
    d = A.versioned(A.DeviceStatus)
    check(lib().gm_device_status(gpu, C.byref(d)), f"device_status({gpu})")
    return d.as_dict()


def device_processes(gpu):
    buf = (A.Process * A.GM_MAX_PROCS)()
    n = C.c_int(A.GM_MAX_PROCS)
    check(lib().gm_device_processes(gpu, buf, C.byref(n)))
    return [_struct_dict(buf[i], skip=()) for i in range(n.value)]


def process_source(gpu):
    """Where device_processes gets its list on this host: "kfd", "fdinfo", "open-files", "fake" or "none"."""
    buf = C.create_string_buffer(32)
    check(lib().gm_process_source(gpu, buf, C.c_size_t(32)))
    return buf.value.decode()


def metrics_table_source(gpu):
    """How the amdsmi backend reads the bulk metrics table of this GPU: "direct" (its own read of the driver's
    gpu_metrics file, checked against the library's decoding), "library" or "library: <why direct reads are off>";
    "n/a" on other backends."""
    buf = C.create_string_buffer(256)
    check(lib().gm_metrics_table_source(gpu, buf, C.c_size_t(256)))
    return buf.value.decode()


def metrics_table_decode(raw):
    """The driver's raw metrics table (bytes of /sys/bus/pci/devices/<bdf>/gpu_metrics) as a dict of the members the
    sampler uses; all-ones = not reported.  Needs no engine.  ValueError for a layout the decoder does not know."""
    import json
    buf = C.create_string_buffer(32768)
    st = lib().gm_metrics_table_decode(bytes(raw), C.c_size_t(len(raw)), buf, C.c_size_t(32768))
    if st == A.GM_ST_NOT_SUPPORTED:
        raise ValueError(buf.value.decode())
    check(st)
    return json.loads(buf.value.decode())


def proc_scan(proc_root="/proc", pci_bdf="", render_minor=-1):
    """The /proc walk on its own: (source, [process dict]) for the card at pci_bdf / render node render_minor."""
    buf = (A.Process * A.GM_MAX_PROCS)()
    n = C.c_int(A.GM_MAX_PROCS)
    src = C.create_string_buffer(32)
    L = lib()
    L.gm_proc_scan.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.POINTER(A.Process), C.POINTER(C.c_int), C.c_char_p, C.c_size_t]
    check(L.gm_proc_scan(_b(proc_root), _b(pci_bdf), render_minor, buf, C.byref(n), src, 32))
    return src.value.decode(), [_struct_dict(buf[i], skip=()) for i in range(n.value)]


def device_mode(gpu):
    m = A.versioned(A.DeviceMode)
    check(lib().gm_device_mode(gpu, C.byref(m)))
    return _struct_dict(m)


# ---- groups / watches ---------------------------------------------------------
def group_create(name, gpus=()):
    g = C.c_uint32()
    check(lib().gm_group_create(_b(name), C.byref(g)), "group_create")
    for i in gpus:
        check(lib().gm_group_add_device(g.value, i), "group_add_device")
    return g.value


def group_add(group, gpu):
    check(lib().gm_group_add_device(group, gpu), "group_add_device")


def group_devices(group):
    ids = (C.c_uint32 * A.GM_MAX_GPUS)()
    n = C.c_int(A.GM_MAX_GPUS)
    check(lib().gm_group_devices(group, ids, C.byref(n)))
    return list(ids[: n.value])


def group_destroy(group):
    check(lib().gm_group_destroy(group), "group_destroy")


def field_group_create(name, field_ids):
    arr = (C.c_uint16 * len(field_ids))(*field_ids)
    fg = C.c_uint32()
    check(lib().gm_field_group_create(arr, len(field_ids), _b(name), C.byref(fg)), "field_group_create")
    return fg.value


def field_group_destroy(fg):
    check(lib().gm_field_group_destroy(fg), "field_group_destroy")


def watch_fields(group, fg, update_freq_us=1_000_000, max_keep_age_s=300.0, max_keep_samples=0):
    check(lib().gm_watch_fields(group, fg, update_freq_us, max_keep_age_s, max_keep_samples), "watch_fields")


def unwatch_fields(group, fg):
    check(lib().gm_unwatch_fields(group, fg), "unwatch_fields")


def update_all_fields(wait=True):
    check(lib().gm_update_all_fields(1 if wait else 0), "update_all_fields")


def latest_values(gpu, field_ids, raw=False):
    arr = (C.c_uint16 * len(field_ids))(*field_ids)
    out = (A.Value * len(field_ids))()
    check(lib().gm_latest_values(gpu, arr, len(field_ids), out), "latest_values")
    if raw:
        return list(out)
    return {fid: out[i].py() for i, fid in enumerate(field_ids)}


def values_since(gpu, field_id, since_ts_us=0, capacity=4096):
    out = (A.Value * capacity)()
    n = C.c_int(capacity)
    st = lib().gm_values_since(gpu, field_id, since_ts_us, out, C.byref(n))
    if st not in (A.GM_ST_OK, A.GM_ST_INSUFFICIENT_SIZE):
        raise GpumonError(st, "values_since")
    return [(out[i].ts_us, out[i].py()) for i in range(min(n.value, capacity))]


def inject_value(gpu, field_id, value, hold_ms=0):
    m = field_meta(field_id)
    v = A.Value()
    v.field_id = field_id
    v.type = m["type"]
    v.status = A.GM_VS_OK
    if m["type"] == A.GM_FT_DOUBLE:
        v.v.dbl = float(value)
    else:
        v.v.i64 = int(value)
    check(lib().gm_inject_value(gpu, C.byref(v), hold_ms), "inject_value")


def fake_ctl(command):
    check(lib().gm_fake_ctl(_b(command)), f"fake_ctl({command})")


def field_meta(field_id):
    m = A.FieldMeta()
    check(lib().gm_field_meta(field_id, C.byref(m)), f"field_meta({field_id})")
    return _struct_dict(m, skip=("reserved",))


def field_list():
    ids = (C.c_uint16 * 1024)()
    n = C.c_int(1024)
    check(lib().gm_field_list(ids, C.byref(n)))
    return list(ids[: n.value])


# ---- health ---------------------------------------------------------------------
def health_set(group, systems=A.HEALTH["ALL"]):
    check(lib().gm_health_set(group, systems), "health_set")


def health_check(group):
    buf = (A.HealthResponse * A.GM_MAX_GPUS)()
    for r in buf:
        r.version = A._version(A.HealthResponse)
    n = C.c_int(A.GM_MAX_GPUS)
    check(lib().gm_health_check(group, buf, C.byref(n)), "health_check")
    out = []
    for i in range(n.value):
        r = buf[i]
        out.append({
            "gpu": r.gpu,
            "status": lib().gm_health_result_name(r.overall).decode(),
            "overall": r.overall,
            "watches": [{"type": lib().gm_health_system_name(r.incidents[k].system).decode(),
                         "system": r.incidents[k].system,
                         "status": lib().gm_health_result_name(r.incidents[k].health).decode(),
                         "health": r.incidents[k].health,
                         "error": r.incidents[k].message.decode()} for k in range(r.incident_count)],
        })
    return out


def health_thresholds_get():
    t = A.versioned(A.HealthThresholds)
    check(lib().gm_health_thresholds_get(C.byref(t)), "health_thresholds_get")
    return _struct_dict(t)


def health_thresholds_set(**levels):
    """Change some of the health-check levels (hotspot_warn_c, hotspot_fail_c, mem_warn_c, mem_fail_c, bad_pages_fail,
    ce_burst_warn, power_over_cap_warn); the others keep their values."""
    t = A.versioned(A.HealthThresholds)
    check(lib().gm_health_thresholds_get(C.byref(t)), "health_thresholds_get")
    for k, v in levels.items():
        if k not in dict(A.HealthThresholds._fields_) or k == "version":
            raise ValueError(f"unknown health threshold {k}")
        setattr(t, k, v)
    check(lib().gm_health_thresholds_set(C.byref(t)), "health_thresholds_set")


def diag_result_submit(gpu, test, system, health, message="", value=float("nan"), ts_us=0):
    """Hand the verdict of an active diagnostic to the engine (see gm_diag_result_t)."""
    d = A.versioned(A.DiagResult)
    d.gpu, d.test, d.system, d.health, d.value, d.ts_us, d.message = gpu, _b(test), system, health, value, ts_us, _b(message)
    check(lib().gm_diag_result_submit(C.byref(d)), "diag_result_submit")


def diag_results(gpu):
    buf = (A.DiagResult * 64)()
    n = C.c_int(64)
    check(lib().gm_diag_results(gpu, buf, C.byref(n)), "diag_results")
    return [_struct_dict(buf[i]) for i in range(n.value)]


def health_check_gpu(gpu):
    """Same shape as the reference's HealthCheckByGpuId (bindings/go/dcgm/health.go:34-86)."""
    g = group_create(f"py-health-{gpu}-{id(object())}", [gpu])
    try:
        health_set(g)
        return health_check(g)[0]
    finally:
        group_destroy(g)


# ---- policy ---------------------------------------------------------------------
def policy_set(group, conditions, max_bad_pages=10, thermal_c=100, power_w=1400, hysteresis_pct=3):
    p = A.versioned(A.Policy)
    p.conditions = conditions
    p.max_bad_pages, p.thermal_c, p.power_w, p.hysteresis_pct = max_bad_pages, thermal_c, power_w, hysteresis_pct
    check(lib().gm_policy_set(group, C.byref(p)), "policy_set")


def policy_get(group):
    p = A.versioned(A.Policy)
    check(lib().gm_policy_get(group, C.byref(p)), "policy_get")
    return _struct_dict(p)


def _violation_dict(v):
    d = {"condition": v.condition, "name": lib().gm_policy_condition_name(v.condition).decode(), "gpu": v.gpu,
         "ts_us": v.ts_us}
    c = v.condition
    if c == A.POLICY["ECC_UE"]:
        d["data"] = {"location": v.d.ecc.location.decode(), "count": v.d.ecc.count}
    elif c == A.POLICY["PCIE"]:
        d["data"] = {"replay_counter": v.d.pcie.replay_counter}
    elif c == A.POLICY["MAX_BAD_PAGES"]:
        d["data"] = {"ce_pages": v.d.bad_pages.ce_pages, "ue_pages": v.d.bad_pages.ue_pages}
    elif c == A.POLICY["THERMAL"]:
        d["data"] = {"temperature_c": v.d.thermal.temperature_c}
    elif c == A.POLICY["POWER"]:
        d["data"] = {"power_w": v.d.power.power_w}
    elif c == A.POLICY["XGMI"]:
        d["data"] = {"field_id": v.d.xgmi.field_id, "link": v.d.xgmi.link, "counter": v.d.xgmi.counter}
    elif c == A.POLICY["GPU_EVENT"]:
        d["data"] = {"code": v.d.event.code, "message": v.d.event.message.decode()}
    return d


class PolicyRegistration:
    """A live policy registration; iterate or call wait() for violations."""

    def __init__(self, group, conditions, callback=None):
        self._cb_keepalive = None
        cb = A.VIOLATION_CB()
        if callback is not None:
            def tramp(vp, _user):
                try:
                    callback(_violation_dict(vp.contents))
                except Exception:  # a Python error must not unwind into the notifier thread
                    pass
                return 0
            cb = A.VIOLATION_CB(tramp)
            self._cb_keepalive = cb
        reg = C.c_uint32()
        check(lib().gm_policy_register(group, conditions, cb, None, C.byref(reg)), "policy_register")
        self.reg = reg.value

    def wait(self, timeout_ms=1000):
        v = A.Violation()
        st = lib().gm_policy_wait(self.reg, timeout_ms, C.byref(v))
        if st == A.GM_ST_TIMEOUT:
            return None
        check(st, "policy_wait")
        return _violation_dict(v)

    def close(self):
        if self.reg:
            lib().gm_policy_unregister(self.reg)
            self.reg = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- accounting / introspection / events ------------------------------------------
def watch_pid_fields(group=A.GM_GROUP_ALL_GPUS, update_freq_us=1_000_000, max_keep_age_s=300.0):
    check(lib().gm_watch_pid_fields(group, update_freq_us, max_keep_age_s, 0), "watch_pid_fields")


def pid_info(pid, group=A.GM_GROUP_ALL_GPUS):
    info = A.versioned(A.PidInfo)
    check(lib().gm_pid_info(group, pid, C.byref(info)), f"pid_info({pid})")
    return [_struct_dict(info.gpus[i], skip=()) | {"critical_event_ts_us": list(info.gpus[i].critical_event_ts_us)[: info.gpus[i].num_critical_events]}
            for i in range(info.num_gpus)]


def introspect():
    lib().gm_introspect_toggle(1)
    s = A.versioned(A.Introspect)
    check(lib().gm_introspect(C.byref(s)), "introspect")
    return _struct_dict(s)


class EventSet:
    def __init__(self, gpu=A.GM_MAX_GPUS, mask=0):
        es = C.c_uint32()
        check(lib().gm_event_set_create(C.byref(es)), "event_set_create")
        self.es = es.value
        check(lib().gm_event_register(self.es, gpu, mask), "event_register")

    def wait(self, timeout_ms=1000):
        e = A.Event()
        st = lib().gm_event_wait(self.es, timeout_ms, C.byref(e))
        if st == A.GM_ST_TIMEOUT:
            return None
        check(st, "event_wait")
        return _struct_dict(e, skip=())

    def close(self):
        if self.es:
            lib().gm_event_set_free(self.es)
            self.es = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


# ---- exporter ---------------------------------------------------------------------
class Exporter:
    def __init__(self, interval_ms=1000, gpus=None, output_path=None, listen=None, profiling=False, extras=False,
                 self_metrics=False, rest_api=False, legacy_layout=False, kubelet_socket=None, pod_map_path=None,
                 resource_names=None, diag_results_path=None, state_path=None, health=False, slow_interval_ms=0,
                 pod_refresh_ms=0):
        c = A.versioned(A.ExporterConfig)
        c.interval_ms = interval_ms
        c.slow_interval_ms, c.pod_refresh_ms = slow_interval_ms, pod_refresh_ms
        c.profiling, c.amd_extras, c.self_metrics = int(profiling), int(extras), int(self_metrics)
        c.rest_api, c.legacy_layout, c.health = int(rest_api), int(legacy_layout), int(health)
        self._keep = [_b(x) for x in (gpus, output_path, listen, kubelet_socket, resource_names, pod_map_path,
                                      diag_results_path, state_path)]
        (c.gpus, c.output_path, c.listen, c.kubelet_socket, c.resource_names, c.pod_map_path,
         c.diag_results_path, c.state_path) = self._keep
        h = C.c_uint32()
        check(lib().gm_exporter_create(C.byref(c), C.byref(h)), "exporter_create")
        self.h = h.value

    def render(self):
        cap = 1 << 20
        while True:
            buf = C.create_string_buffer(cap)
            n = C.c_size_t(cap)
            st = lib().gm_exporter_render(self.h, buf, C.byref(n))
            if st == A.GM_ST_INSUFFICIENT_SIZE:
                cap = n.value + 1024
                continue
            check(st, "exporter_render")
            return buf.value.decode()

    def sweep(self):
        check(lib().gm_exporter_sweep(self.h), "exporter_sweep")

    def start(self):
        check(lib().gm_exporter_start(self.h), "exporter_start")

    def stop(self):
        lib().gm_exporter_stop(self.h)

    @property
    def port(self):
        p = C.c_int()
        check(lib().gm_exporter_http_port(self.h, C.byref(p)))
        return p.value

    def close(self):
        if self.h:
            lib().gm_exporter_destroy(self.h)
            self.h = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def render_view(view, ident=0, json=False):
    views = {"device_info": 0, "device_status": 1, "health": 2, "process": 3, "hostengine": 4, "topology": 5, "links": 6, "diag": 7}
    cap = 1 << 16
    while True:
        buf = C.create_string_buffer(cap)
        n = C.c_size_t(cap)
        st = lib().gm_render_view(views[view], ident, int(json), buf, C.byref(n))
        if st == A.GM_ST_INSUFFICIENT_SIZE:
            cap = n.value + 1024
            continue
        check(st, f"render_view({view})")
        return buf.value.decode()


def pod_relabel(text, pod_map, index_prefix="amdgpu"):
    """pod_map: {device_id: (pod, namespace, container)}."""
    tsv = "".join(f"{k}\t{v[0]}\t{v[1]}\t{v[2]}\n" for k, v in pod_map.items())
    cap = len(text) * 2 + 4096
    buf = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    check(lib().gm_pod_relabel(_b(text), _b(tsv), _b(index_prefix), buf, C.byref(n)), "pod_relabel")
    return buf.value.decode()


def kubelet_pod_map(socket_path, resource_names="amd.com/gpu", timeout_ms=10000):
    cap = 1 << 20
    buf = C.create_string_buffer(cap)
    n = C.c_size_t(cap)
    check(lib().gm_kubelet_pod_map(_b(socket_path), _b(resource_names), timeout_ms, buf, C.byref(n)), "kubelet_pod_map")
    out = {}
    for line in buf.value.decode().splitlines():
        f = line.split("\t")
        if len(f) >= 4:
            out[f[0]] = (f[1], f[2], f[3])
    return out
