"""Locates and loads the native libraries (libgpumon.so, libgpumon_diag.so).

The libraries are built in-tree under ``build/`` by ``make`` (see the repo's
Makefile or ``__graft_entry__.build``).  Nothing here falls back to a Python
re-implementation: a missing library is an ImportError-style failure.
"""
import ctypes as C
import os
import subprocess

from . import _abi as A

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD_DIR = os.environ.get("GPUMON_BUILD_DIR", os.path.join(REPO_ROOT, "build"))


class GpumonError(RuntimeError):
    def __init__(self, status, what=""):
        self.status = status
        msg = lib().gm_strerror(status).decode() if _lib is not None else str(status)
        super().__init__(f"{what}: {msg} ({status})" if what else f"{msg} ({status})")


_lib = None
_diag = None


def build(targets=("all",), jobs=8):
    """Runs make for the given targets in the repo root."""
    subprocess.run(["make", f"-j{jobs}", *targets], cwd=REPO_ROOT, check=True,
                   stdout=subprocess.DEVNULL if not os.environ.get("GPUMON_VERBOSE_BUILD") else None)


def _find(name):
    env = os.environ.get("GPUMON_LIB_DIR")
    for d in ([env] if env else []) + [BUILD_DIR]:
        p = os.path.join(d, name)
        if os.path.exists(p):
            return p
    raise OSError(f"{name} not found under {BUILD_DIR}; run `make` (or __graft_entry__.build()) first")


def lib():
    """libgpumon.so with argument / result types declared."""
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(_find("libgpumon.so"), mode=C.RTLD_GLOBAL)
    st = C.c_int
    u32p = C.POINTER(C.c_uint32)
    sig = {
        "gm_strerror": (C.c_char_p, [C.c_int]),
        "gm_init": (st, [C.POINTER(A.InitParams)]),
        "gm_shutdown": (st, []),
        "gm_init_refcount": (C.c_int, []),
        "gm_server_start": (st, [C.c_char_p, C.c_int]),
        "gm_server_stop": (st, []),
        "gm_field_meta": (st, [C.c_uint16, C.POINTER(A.FieldMeta)]),
        "gm_field_by_tag": (st, [C.c_char_p, C.POINTER(C.c_uint16)]),
        "gm_field_list": (st, [C.POINTER(C.c_uint16), C.POINTER(C.c_int)]),
        "gm_device_count": (st, [u32p]),
        "gm_supported_devices": (st, [u32p, C.POINTER(C.c_int)]),
        "gm_device_attributes": (st, [C.c_uint32, C.POINTER(A.DeviceAttributes)]),
        "gm_device_topology": (st, [C.c_uint32, C.POINTER(A.Topology)]),
        "gm_xgmi_links": (st, [C.c_uint32, C.POINTER(A.XgmiLink), C.POINTER(C.c_int)]),
        "gm_health_thresholds_set": (st, [C.POINTER(A.HealthThresholds)]),
        "gm_health_thresholds_get": (st, [C.POINTER(A.HealthThresholds)]),
        "gm_diag_result_submit": (st, [C.POINTER(A.DiagResult)]),
        "gm_diag_results": (st, [C.c_uint32, C.POINTER(A.DiagResult), C.POINTER(C.c_int)]),
        "gm_link_type_name": (C.c_char_p, [C.c_int32, C.c_uint32]),
        "gm_link_type_description": (C.c_char_p, [C.c_int32]),
        "gm_group_create": (st, [C.c_char_p, u32p]),
        "gm_group_add_device": (st, [C.c_uint32, C.c_uint32]),
        "gm_group_remove_device": (st, [C.c_uint32, C.c_uint32]),
        "gm_group_devices": (st, [C.c_uint32, u32p, C.POINTER(C.c_int)]),
        "gm_group_destroy": (st, [C.c_uint32]),
        "gm_field_group_create": (st, [C.POINTER(C.c_uint16), C.c_int, C.c_char_p, u32p]),
        "gm_field_group_destroy": (st, [C.c_uint32]),
        "gm_watch_fields": (st, [C.c_uint32, C.c_uint32, C.c_int64, C.c_double, C.c_int32]),
        "gm_unwatch_fields": (st, [C.c_uint32, C.c_uint32]),
        "gm_update_all_fields": (st, [C.c_int]),
        "gm_latest_values": (st, [C.c_uint32, C.POINTER(C.c_uint16), C.c_int, C.POINTER(A.Value)]),
        "gm_values_since": (st, [C.c_uint32, C.c_uint16, C.c_int64, C.POINTER(A.Value), C.POINTER(C.c_int)]),
        "gm_inject_value": (st, [C.c_uint32, C.POINTER(A.Value), C.c_int32]),
        "gm_fake_ctl": (st, [C.c_char_p]),
        "gm_device_status": (st, [C.c_uint32, C.POINTER(A.DeviceStatus)]),
        "gm_device_processes": (st, [C.c_uint32, C.POINTER(A.Process), C.POINTER(C.c_int)]),
        "gm_device_mode": (st, [C.c_uint32, C.POINTER(A.DeviceMode)]),
        "gm_health_set": (st, [C.c_uint32, C.c_uint32]),
        "gm_health_get": (st, [C.c_uint32, u32p]),
        "gm_health_check": (st, [C.c_uint32, C.POINTER(A.HealthResponse), C.POINTER(C.c_int)]),
        "gm_health_system_name": (C.c_char_p, [C.c_uint32]),
        "gm_health_result_name": (C.c_char_p, [C.c_int32]),
        "gm_policy_set": (st, [C.c_uint32, C.POINTER(A.Policy)]),
        "gm_policy_get": (st, [C.c_uint32, C.POINTER(A.Policy)]),
        "gm_policy_register": (st, [C.c_uint32, C.c_uint32, A.VIOLATION_CB, C.c_void_p, u32p]),
        "gm_policy_unregister": (st, [C.c_uint32]),
        "gm_policy_wait": (st, [C.c_uint32, C.c_int32, C.POINTER(A.Violation)]),
        "gm_policy_condition_name": (C.c_char_p, [C.c_uint32]),
        "gm_watch_pid_fields": (st, [C.c_uint32, C.c_int64, C.c_double, C.c_int32]),
        "gm_pid_info": (st, [C.c_uint32, C.c_uint32, C.POINTER(A.PidInfo)]),
        "gm_introspect_toggle": (st, [C.c_int]),
        "gm_introspect": (st, [C.POINTER(A.Introspect)]),
        "gm_event_set_create": (st, [u32p]),
        "gm_event_register": (st, [C.c_uint32, C.c_uint32, C.c_uint64]),
        "gm_event_wait": (st, [C.c_uint32, C.c_int32, C.POINTER(A.Event)]),
        "gm_event_set_free": (st, [C.c_uint32]),
        "gm_exporter_create": (st, [C.POINTER(A.ExporterConfig), u32p]),
        "gm_exporter_render": (st, [C.c_uint32, C.c_char_p, C.POINTER(C.c_size_t)]),
        "gm_exporter_sweep": (st, [C.c_uint32]),
        "gm_exporter_start": (st, [C.c_uint32]),
        "gm_exporter_stop": (st, [C.c_uint32]),
        "gm_exporter_http_port": (st, [C.c_uint32, C.POINTER(C.c_int)]),
        "gm_exporter_destroy": (st, [C.c_uint32]),
        "gm_pod_relabel": (st, [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_size_t)]),
        "gm_kubelet_pod_map": (st, [C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.POINTER(C.c_size_t)]),
        "gm_pod_exporter_start": (st, [C.POINTER(A.PodExporterConfig), u32p]),
        "gm_pod_exporter_status": (st, [C.c_uint32, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_uint64)]),
        "gm_pod_exporter_stop": (st, [C.c_uint32]),
        "gm_render_view": (st, [C.c_int, C.c_uint32, C.c_int, C.c_char_p, C.POINTER(C.c_size_t)]),
        "gm_format_violation": (st, [C.POINTER(A.Violation), C.c_char_p, C.POINTER(C.c_size_t)]),
        "gm_event_code_name": (C.c_char_p, [C.c_uint32]),
        "gm_set_log_level": (None, [C.c_int]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(L, name)
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def check(status, what=""):
    if status != A.GM_ST_OK:
        raise GpumonError(status, what)


def diag_lib():
    """libgpumon_diag.so (HIP kernels).  Import torch first when both are used in one process."""
    global _diag
    if _diag is None:
        _diag = C.CDLL(_find("libgpumon_diag.so"), mode=C.RTLD_GLOBAL)
    return _diag


def binary(name):
    p = os.path.join(BUILD_DIR, "bin", name)
    if not os.path.exists(p):
        raise OSError(f"{p} not built")
    return p
