"""Device-wide hardware counters (build/libgpumon_prof.so, core/prof/prof_counters.h): MFMA pipe
activity, graphics-engine / shader-array / memory-pipeline activity and the shader clock held, per window,
for work of ANY process.

The plugin registers with rocprofiler-sdk and opens the compute runtime (HSA) itself.  Use it from a
monitoring process that does not run HIP work of its own (what ``gpumon-exporter --prof-counters``
does): once HSA is running :func:`open` is refused, and after :func:`open` the HIP runtime of the same
process no longer finds a device (measured with torch on ROCm 7.2).  This is the source
of ``dcgm_fi_prof_pipe_tensor_active``, the analogue of the reference's ``-p`` tensor-pipe metric
(exporters/prometheus-dcgm/dcgm-exporter/dcgm-exporter:80-83,179-187).
"""
import ctypes as C
import math

from .._lib import _find


class ProfError(RuntimeError):
    pass


class _AgentInfo(C.Structure):
    _fields_ = [("pci_domain", C.c_uint32), ("pci_location", C.c_uint32), ("cu_count", C.c_uint32), ("simd_count", C.c_uint32),
                ("name", C.c_char * 64)]


class _Sample(C.Structure):
    _fields_ = [("window_s", C.c_double), ("gpu_active", C.c_double), ("mfma_active", C.c_double), ("clock_mhz", C.c_double),
                ("cu_active", C.c_double), ("mem_pipe_active", C.c_double), ("grbm_busy", C.c_double), ("grbm_name", C.c_char * 32),
                ("grbm_instances", C.c_int32), ("reserved", C.c_int32), ("raw", C.c_double * 8)]


ABI_VERSION = 2


_L = None


def _lib():
    global _L
    if _L is None:
        L = C.CDLL(_find("libgpumon_prof.so"), mode=C.RTLD_LOCAL)
        if not hasattr(L, "gmp_abi_version") or L.gmp_abi_version() != ABI_VERSION:
            raise ProfError("libgpumon_prof.so was built for another layout of gmp_sample_t: rebuild it (make prof)")
        L.gmp_open.argtypes, L.gmp_open.restype = [C.c_char_p, C.c_int], C.c_int
        L.gmp_set_count.argtypes, L.gmp_set_count.restype = [], C.c_int
        L.gmp_agent_info.argtypes, L.gmp_agent_info.restype = [C.c_int, C.POINTER(_AgentInfo)], C.c_int
        L.gmp_sample.argtypes, L.gmp_sample.restype = [C.c_int, C.POINTER(_Sample)], C.c_int
        L.gmp_last_error.argtypes, L.gmp_last_error.restype = [], C.c_char_p
        L.gmp_close.argtypes, L.gmp_close.restype = [], None
        _L = L
    return _L


def open():  # noqa: A001 - mirrors gmp_open
    """Registers, opens HSA, sets up the counters.  Returns the list of GPU agents (dicts)."""
    err = C.create_string_buffer(512)
    n = _lib().gmp_open(err, len(err))
    if n < 0:
        raise ProfError(err.value.decode(errors="replace") or "gmp_open failed")
    agents = []
    for i in range(n):
        ai = _AgentInfo()
        if _lib().gmp_agent_info(i, C.byref(ai)) != 0:
            raise ProfError(f"gmp_agent_info({i}) failed")
        loc = ai.pci_location
        agents.append({"agent": i, "name": ai.name.decode(), "cu_count": ai.cu_count, "simd_count": ai.simd_count,
                       "pci_bdf": f"{ai.pci_domain:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"})
    return agents


def set_count():
    """Counter sets the windows rotate through: after this many samples every metric has been measured once."""
    return _lib().gmp_set_count()


def sample(agent=0):
    """One window: None on the first call for an agent (it only starts the counters), then a dict with
    ``window_s``, ``mfma_active`` (percent of all cycles of the window), ``clock_mhz`` and ONE of ``gpu_active`` /
    ``cu_active`` / ``mem_pipe_active``: the GRBM block has two counter slots, one holds the clock, the other rotates
    from window to window (``grbm_name`` says which it was).  A value the window did not measure or the device
    cannot give is None, never 0."""
    s = _Sample()
    rc = _lib().gmp_sample(agent, C.byref(s))
    if rc < 0:
        raise ProfError(_lib().gmp_last_error().decode(errors="replace"))
    if rc == 1:
        return None
    opt = lambda v: None if math.isnan(v) else v  # noqa: E731
    name = s.grbm_name.decode()
    return {"window_s": s.window_s, "gpu_active": opt(s.gpu_active), "cu_active": opt(s.cu_active),
            "mem_pipe_active": opt(s.mem_pipe_active), "mfma_active": opt(s.mfma_active), "clock_mhz": opt(s.clock_mhz),
            "grbm_name": name, "grbm_busy": opt(s.grbm_busy), "grbm_instances": s.grbm_instances,
            "raw": {"GRBM_COUNT": s.raw[0], name: s.raw[1], "SQ_VALU_MFMA_BUSY_CYCLES": s.raw[2]}}


def sample_all(agent=0, window_s=0.25, sleep=None):
    """One window per counter set, merged: every metric the device can give, each from its own window."""
    import time
    sleep = sleep or time.sleep
    merged, mfma = {}, []
    if sample(agent) is not None:
        pass  # counters were already running: that window's length is unknown to the caller, drop it
    for _ in range(max(1, set_count())):
        sleep(window_s)
        w = sample(agent)
        if w is None:
            continue
        for k in ("gpu_active", "cu_active", "mem_pipe_active", "clock_mhz"):
            if w[k] is not None:
                merged[k] = w[k]
        if w["mfma_active"] is not None:
            mfma.append(w["mfma_active"])
    merged["mfma_active"] = sum(mfma) / len(mfma) if mfma else None
    for k in ("gpu_active", "cu_active", "mem_pipe_active", "clock_mhz"):
        merged.setdefault(k, None)
    return merged


def close():
    """Stops the counters (the runtime stays open for the life of the process)."""
    _lib().gmp_close()
