"""ctypes front end of libgpumon_diag.so — the hand-written CDNA4 diagnostics.

No fallback path: on a machine with a GPU the native library must load, and
every function here runs the HIP kernels of diag/hip (never a torch stand-in).
torch is only ever used by callers as an *oracle* to check numerics.
"""
import ctypes as C

from .._lib import diag_lib

STREAM_KERNELS = {"copy": 0, "scale": 1, "add": 2, "triad": 3, "read": 4, "write": 5}
PATTERNS = {"walking_ones": 1, "walking_zeros": 2, "address": 4, "checkerboard": 8, "inv_checkerboard": 16,
            "random": 32, "modulo_20": 64}
PAT_ALL = 0x7F
DTYPES = {"bf16": 0, "fp8": 1, "f32": 2, "mxfp8": 3}
MEMTEST_LOG = 32


class DiagError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        msg = diag_lib().gmd_last_error().decode()
        super().__init__(f"{what} failed ({code}): {msg}")


class DeviceInfo(C.Structure):
    _fields_ = [("name", C.c_char * 128), ("gcn_arch", C.c_char * 64), ("total_mem", C.c_uint64),
                ("free_mem", C.c_uint64), ("compute_units", C.c_int), ("clock_mhz", C.c_int),
                ("mem_clock_mhz", C.c_int), ("l2_bytes", C.c_int), ("lds_per_cu", C.c_int), ("pci_bdf", C.c_char * 32)]


class StreamResult(C.Structure):
    _fields_ = [("tbps_best", C.c_double), ("tbps_median", C.c_double), ("ms_best", C.c_double),
                ("bytes_per_iter", C.c_uint64), ("verified", C.c_int32), ("iters", C.c_int32)]


class MemtestError(C.Structure):
    _fields_ = [("byte_offset", C.c_uint64), ("expected", C.c_uint64), ("actual", C.c_uint64),
                ("pattern", C.c_uint32), ("chunk", C.c_uint32), ("device_address", C.c_uint64),
                ("flipped_bits", C.c_uint32), ("page_offset", C.c_uint32)]


class MemtestResult(C.Structure):
    _fields_ = [("bytes_tested", C.c_uint64), ("bytes_requested", C.c_uint64), ("errors", C.c_uint64),
                ("patterns_run", C.c_uint32), ("chunks", C.c_uint32), ("write_tbps", C.c_double),
                ("verify_tbps", C.c_double), ("seconds", C.c_double), ("logged", C.c_uint32),
                ("log", MemtestError * MEMTEST_LOG), ("scrubbed", C.c_uint32), ("march_passes", C.c_uint32),
                ("march_tbps", C.c_double)]


PROGRESS_CB = C.CFUNCTYPE(None, C.c_double, C.c_char_p, C.c_void_p)


class MemtestConfig(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("fraction", C.c_double), ("chunk_bytes", C.c_uint64),
                ("patterns", C.c_uint32), ("passes", C.c_uint32), ("seed", C.c_uint64),
                ("inject_offset", C.c_uint64), ("inject_xor", C.c_uint64), ("progress", PROGRESS_CB),
                ("progress_user", C.c_void_p), ("march", C.c_uint32)]


class MfmaResult(C.Structure):
    _fields_ = [("tflops", C.c_double), ("tflops_best", C.c_double), ("seconds", C.c_double),
                ("launches", C.c_uint64), ("clock_mhz_est", C.c_double), ("verified", C.c_int32),
                ("sdc_mismatches", C.c_uint64), ("max_rel_err", C.c_double), ("checked_in_kernel", C.c_uint64),
                ("checked_in_memory", C.c_uint64)]


_typed = False


def _lib():
    global _typed
    L = diag_lib()
    if not _typed:
        L.gmd_last_error.restype = C.c_char_p
        L.gmd_device_count.restype = C.c_int
        L.gmd_device_info.argtypes = [C.c_int, C.POINTER(DeviceInfo)]
        L.gmd_hbm_stream.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(StreamResult)]
        L.gmd_hbm_stream_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64,
                                            C.c_int, C.c_void_p]
        L.gmd_memtest.argtypes = [C.c_int, C.POINTER(MemtestConfig), C.POINTER(MemtestResult)]
        L.gmd_memtest_fill.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p]
        L.gmd_memtest_verify.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint64,
                                         C.POINTER(C.c_uint64), C.POINTER(MemtestError), C.POINTER(C.c_uint32),
                                         C.c_void_p]
        L.gmd_mfma_peak.argtypes = [C.c_int, C.c_int, C.c_double, C.POINTER(MfmaResult)]
        L.gmd_gemm_stress.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int,
                                      C.POINTER(MfmaResult)]
        L.gmd_gemm_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                      C.c_void_p]
        L.gmd_gemm_launch_mx.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                         C.c_void_p]
        L.gmd_mx_pack_scales.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        _typed = True
    return L


def _dict(s):
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        if isinstance(v, bytes):
            v = v.decode()
        if isinstance(v, C.Array):
            continue
        out[name] = v
    return out


def device_count():
    return _lib().gmd_device_count()


def device_info(device=0):
    d = DeviceInfo()
    rc = _lib().gmd_device_info(device, C.byref(d))
    if rc:
        raise DiagError(rc, "gmd_device_info")
    return _dict(d)


def _nt(nontemporal):
    """False -> plain, True -> non-temporal, "auto" -> the faster measured policy per kernel."""
    return 2 if nontemporal == "auto" else int(bool(nontemporal))


def hbm_stream(kernel="copy", bytes_per_buffer=4 << 30, iters=10, nontemporal="auto", device=0):
    r = StreamResult()
    rc = _lib().gmd_hbm_stream(device, STREAM_KERNELS[kernel], bytes_per_buffer, iters, _nt(nontemporal), C.byref(r))
    if rc:
        raise DiagError(rc, f"gmd_hbm_stream({kernel})")
    return _dict(r)


def hbm_stream_launch(kernel, a, b, c, scalar, n_floats, nontemporal=False, stream=None):
    """One launch on caller memory (device pointers as ints, e.g. tensor.data_ptr())."""
    rc = _lib().gmd_hbm_stream_launch(STREAM_KERNELS[kernel], a, b, c, scalar, n_floats, _nt(nontemporal), stream)
    if rc:
        raise DiagError(rc, f"gmd_hbm_stream_launch({kernel})")


def memtest(bytes=0, fraction=0.90, chunk_bytes=2 << 30, patterns=PAT_ALL, passes=1, seed=0, inject_offset=0,
            inject_xor=0, progress=None, device=0, march=False):
    cfg = MemtestConfig()
    cfg.march = int(march)
    cfg.bytes, cfg.fraction, cfg.chunk_bytes = bytes, fraction, chunk_bytes
    cfg.patterns, cfg.passes, cfg.seed = patterns, passes, seed
    cfg.inject_offset, cfg.inject_xor = inject_offset, inject_xor
    keep = None
    if progress is not None:
        keep = PROGRESS_CB(lambda f, what, _u: progress(f, what.decode()))
        cfg.progress = keep
    r = MemtestResult()
    rc = _lib().gmd_memtest(device, C.byref(cfg), C.byref(r))
    if rc:
        raise DiagError(rc, "gmd_memtest")
    d = _dict(r)
    d["log"] = [_dict(r.log[i]) for i in range(r.logged)]
    return d


def memtest_fill(ptr, n_words, pattern, seed=0, base_word=0, stream=None):
    rc = _lib().gmd_memtest_fill(ptr, n_words, base_word, PATTERNS[pattern] if isinstance(pattern, str) else pattern,
                                 seed, stream)
    if rc:
        raise DiagError(rc, "gmd_memtest_fill")


def memtest_verify(ptr, n_words, pattern, seed=0, base_word=0, stream=None):
    errs = C.c_uint64()
    log = (MemtestError * MEMTEST_LOG)()
    n = C.c_uint32()
    rc = _lib().gmd_memtest_verify(ptr, n_words, base_word, PATTERNS[pattern] if isinstance(pattern, str) else pattern,
                                   seed, C.byref(errs), log, C.byref(n), stream)
    if rc:
        raise DiagError(rc, "gmd_memtest_verify")
    return errs.value, [_dict(log[i]) for i in range(n.value)]


def mfma_peak(dtype="bf16", seconds=1.0, device=0):
    r = MfmaResult()
    rc = _lib().gmd_mfma_peak(device, DTYPES[dtype], seconds, C.byref(r))
    if rc:
        raise DiagError(rc, f"gmd_mfma_peak({dtype})")
    return _dict(r)


class GemmStressConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("dtype", C.c_int32), ("m", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("verify", C.c_int32), ("check_every", C.c_int32), ("seconds", C.c_double)]


def gemm_stress(dtype="bf16", m=8192, n=8192, k=8192, seconds=2.0, verify=True, device=0, check_every=0):
    """GEMM soak: back-to-back launches, the first compared with an fp32 reference, the checksum of every
    `check_every`-th result (0 = the default, 8) compared with the first one's."""
    L = _lib()
    L.gmd_gemm_stress_run.argtypes = [C.POINTER(GemmStressConfig), C.POINTER(MfmaResult)]
    cfg = GemmStressConfig(device, DTYPES[dtype], m, n, k, int(verify), check_every, seconds)
    r = MfmaResult()
    rc = L.gmd_gemm_stress_run(C.byref(cfg), C.byref(r))
    if rc:
        raise DiagError(rc, f"gmd_gemm_stress_run({dtype})")
    return _dict(r)


def gemm_set_variant(variant):
    """-1 = fastest schedule per dtype and shape (default), 0 = simple double-buffered loop (A/B baseline),
    1 = eight waves x 128x64 per tile, 2 = four waves x 128x128 (accumulators in AGPRs), 3 = eight waves x 128x64
    as one stream per CU over its tiles, 4 / 5 = schedule 1 with the narrow / cached epilogue, 6 = schedule 2 as one
    stream per CU (A/B reference; slower than 2, see docs/diagnostics.md), 7 = schedule 2 on the K=32 fp8 instruction
    (unscaled fp8 runs -1, 2 and 6 on the K=128 instruction, at twice the rate, and the others on the K=32 one),
    8 = 128 x 128 tiles, 9 / 10 = small / large tiles split along K (see gemm_plan)."""
    rc = _lib().gmd_gemm_set_variant(int(variant))
    if rc:
        raise DiagError(rc, "gmd_gemm_set_variant")


def gemm_plan(dtype, m, n, k):
    """What the schedule in force does with a GEMM of this shape on the current device: (9 or 10, number of K ranges)
    where it runs split along K (small / large tiles), (0, 1) where it does not.  A split GEMM adds its products up
    in another order than the unsplit schedules, the same one at every launch."""
    schedule, splits = C.c_int(0), C.c_int(1)
    rc = _lib().gmd_gemm_plan(DTYPES[dtype], int(m), int(n), int(k), C.byref(schedule), C.byref(splits))
    if rc:
        raise DiagError(rc, "gmd_gemm_plan")
    return schedule.value, splits.value


def gemm_split_k(dtype, m, n, k):
    """Number of K ranges of gemm_plan() alone (1 = no split)."""
    return gemm_plan(dtype, m, n, k)[1]


def gemm_set_split(splits):
    """Number of K ranges schedules 9 and 10 use when asked for (2..8); 0 = as many as fit."""
    rc = _lib().gmd_gemm_set_split(int(splits))
    if rc:
        raise DiagError(rc, "gmd_gemm_set_split")


COLLECTIVES = {"all_reduce": 0, "all_gather": 1, "reduce_scatter": 2, "all_to_all": 3, "broadcast": 4}


class RcclConfig(C.Structure):
    _fields_ = [("ndev", C.c_int32), ("devices", C.POINTER(C.c_int)), ("ops_mask", C.c_uint32),
                ("min_bytes", C.c_uint64), ("max_bytes", C.c_uint64), ("factor", C.c_int32), ("iters", C.c_int32),
                ("warmup", C.c_int32), ("verify", C.c_int32)]


class RcclPoint(C.Structure):
    _fields_ = [("op", C.c_int32), ("ndev", C.c_int32), ("bytes", C.c_uint64), ("iters", C.c_int32),
                ("verified", C.c_int32), ("time_us", C.c_double), ("algbw_gbps", C.c_double),
                ("busbw_gbps", C.c_double)]


def busbw_factor(op, ndev):
    L = _lib()
    L.gmd_busbw_factor.restype = C.c_double
    L.gmd_busbw_factor.argtypes = [C.c_int, C.c_int]
    return L.gmd_busbw_factor(COLLECTIVES[op], ndev)


def rccl_sweep(devices=None, ops=None, min_bytes=8, max_bytes=1 << 30, factor=4, iters=0, warmup=3, verify=True):
    """Single-process RCCL sweep over `devices` (default: all visible); list of dicts, one per (op, size)."""
    L = _lib()
    L.gmd_rccl_sweep.argtypes = [C.POINTER(RcclConfig), C.POINTER(RcclPoint), C.POINTER(C.c_int)]
    cfg = RcclConfig()
    keep = None
    if devices:
        keep = (C.c_int * len(devices))(*devices)
        cfg.ndev, cfg.devices = len(devices), keep
    for o in ops or ():
        cfg.ops_mask |= 1 << COLLECTIVES[o]
    cfg.min_bytes, cfg.max_bytes, cfg.factor = min_bytes, max_bytes, factor
    cfg.iters, cfg.warmup, cfg.verify = iters, warmup, int(verify)
    pts = (RcclPoint * 512)()
    n = C.c_int(512)
    rc = L.gmd_rccl_sweep(C.byref(cfg), pts, C.byref(n))
    if rc:
        raise DiagError(rc, "gmd_rccl_sweep")
    names = {v: k for k, v in COLLECTIVES.items()}
    out = []
    for i in range(n.value):
        d = _dict(pts[i])
        d["op"] = names[d["op"]]
        out.append(d)
    return out


def p2p_matrix(devices=None, bytes=256 << 20, iters=5):
    L = _lib()
    L.gmd_p2p_matrix.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_uint64, C.c_int, C.POINTER(C.c_double)]
    n = len(devices) if devices else device_count()
    arr = (C.c_int * n)(*devices) if devices else None
    m = (C.c_double * (n * n))()
    rc = L.gmd_p2p_matrix(n, arr, bytes, iters, m)
    if rc:
        raise DiagError(rc, "gmd_p2p_matrix")
    return [[m[i * n + j] for j in range(n)] for i in range(n)]


LINK_MODES = {"uni": 0, "bidi": 1, "all": 2}


LINK_MAX_PEERS = 63


class LinkConfig(C.Structure):
    _fields_ = [("ndev", C.c_int32), ("devices", C.POINTER(C.c_int)), ("bytes", C.c_uint64), ("iters", C.c_int32),
                ("verify", C.c_int32), ("workgroups", C.c_int32), ("pull", C.c_int32),
                ("per_gpu_gbps", C.POINTER(C.c_double))]


class LinkRule(C.Structure):
    _fields_ = [("floor_gbps", C.c_double), ("warn_ratio", C.c_double), ("fail_ratio", C.c_double)]


class LinkVerdict(C.Structure):
    _fields_ = [("src", C.c_int32), ("dst", C.c_int32), ("health", C.c_int32), ("gbps", C.c_double),
                ("sibling_median_gbps", C.c_double), ("message", C.c_char * 160)]


class LinkFanResult(C.Structure):
    _fields_ = [("npeers", C.c_int32), ("launch_ms", C.c_double), ("window_ms_min", C.c_double),
                ("window_ms_max", C.c_double), ("launch_gbps", C.c_double), ("tick_hz", C.c_double)]


class LinkExpect(C.Structure):
    _fields_ = [("link_gbps", C.c_double), ("derate_uni", C.c_double), ("derate_bidi", C.c_double),
                ("derate_all", C.c_double), ("pull_factor", C.c_double), ("derate_collective", C.c_double),
                ("warn_ratio", C.c_double), ("fail_ratio", C.c_double), ("scale", C.c_double), ("pull", C.c_int32)]


class LinkFinding(C.Structure):
    _fields_ = [("kind", C.c_int32), ("mode", C.c_int32), ("src", C.c_int32), ("dst", C.c_int32),
                ("health", C.c_int32), ("gbps", C.c_double), ("expected_gbps", C.c_double),
                ("message", C.c_char * 192)]


def link_test(mode="uni", devices=None, bytes=256 << 20, iters=5, verify=True, workgroups=0, pull=False,
              with_per_gpu=False):
    """Peer-copy bandwidth matrix over the xGMI links: a copy kernel on the initiating GPU, one launch per GPU that
    serves all of its peers of the pattern step, each pair timed inside it by its own workgroups' stamps.
    mode: "uni" one pair at a time, "bidi" both directions of a pair at once, "all" every GPU on every peer at
    once.  pull=False: remote stores; pull=True: remote loads.  Returns (gbps, bad): gbps[i][j] = rate at which
    devices[i] wrote to / read from devices[j] in GB/s (-1 without peer access), bad[i][j] = corrupt 16-byte words;
    with_per_gpu adds the per-GPU totals by the HIP events around each launch ("all" only).  One GPU gives
    ([[0.0]], [[0]])."""
    L = _lib()
    L.gmd_link_test.argtypes = [C.POINTER(LinkConfig), C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    n = len(devices) if devices else device_count()
    arr = (C.c_int * n)(*devices) if devices else None
    per = (C.c_double * n)()
    cfg = LinkConfig(n, arr, bytes, iters, 1 if verify else 0, workgroups, 1 if pull else 0, per)
    m, b = (C.c_double * (n * n))(), (C.c_uint64 * (n * n))()
    rc = L.gmd_link_test(C.byref(cfg), LINK_MODES[mode], m, b)
    if rc:
        raise DiagError(rc, f"gmd_link_test({mode})")
    out = ([[m[i * n + j] for j in range(n)] for i in range(n)], [[b[i * n + j] for j in range(n)] for i in range(n)])
    return out + (list(per),) if with_per_gpu else out


def link_selftest(bytes=256 << 20, workgroups=0, device=0):
    """The link test's copy + check kernels inside one GPU: (gbps, bad_words)."""
    L = _lib()
    L.gmd_link_selftest.argtypes = [C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    g, b = C.c_double(), C.c_uint64()
    rc = L.gmd_link_selftest(device, bytes, workgroups, C.byref(g), C.byref(b))
    if rc:
        raise DiagError(rc, "gmd_link_selftest")
    return g.value, b.value


def link_fan_selftest(npeers=7, bytes=64 << 20, workgroups=0, iters=5, device=0):
    """The fused all-peers launch with `npeers` local destinations standing in for the peers: per-destination rates
    from the in-kernel stamps, bad words per destination, and the relation of the stamp windows to the HIP-event
    time of the launch.  Returns (gbps list, bad list, result dict)."""
    L = _lib()
    L.gmd_link_fan_selftest.argtypes = [C.c_int, C.c_int, C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_double),
                                        C.POINTER(C.c_uint64), C.POINTER(LinkFanResult)]
    g, b, r = (C.c_double * npeers)(), (C.c_uint64 * npeers)(), LinkFanResult()
    rc = L.gmd_link_fan_selftest(device, npeers, bytes, workgroups, iters, g, b, C.byref(r))
    if rc:
        raise DiagError(rc, "gmd_link_fan_selftest")
    return list(g), list(b), _dict(r)


class PcieConfig(C.Structure):
    _fields_ = [("bytes", C.c_uint64), ("iters", C.c_int32), ("verify", C.c_int32), ("workgroups", C.c_int32), ("reserved", C.c_int32)]


class PcieResult(C.Structure):
    _fields_ = [("h2d_gbps", C.c_double), ("d2h_gbps", C.c_double), ("bidir_h2d_gbps", C.c_double), ("bidir_d2h_gbps", C.c_double),
                ("kernel_read_gbps", C.c_double), ("kernel_write_gbps", C.c_double), ("duplex_read_gbps", C.c_double), ("duplex_write_gbps", C.c_double),
                ("small_h2d_us", C.c_double), ("small_d2h_us", C.c_double),
                ("bad_h2d", C.c_uint64), ("bad_d2h", C.c_uint64), ("bad_kernel_read", C.c_uint64), ("bad_kernel_write", C.c_uint64),
                ("bytes", C.c_uint64), ("iters", C.c_int32), ("verified", C.c_int32)]


def pcie_test(bytes=256 << 20, iters=5, verify=True, workgroups=0, device=0):
    """The host link of one GPU over pinned memory: copy engines up, down and both at once, a kernel reading and a
    kernel writing host memory; every transfer verified on the far side.  GB/s of the best iteration."""
    L = _lib()
    L.gmd_pcie_test.argtypes = [C.c_int, C.POINTER(PcieConfig), C.POINTER(PcieResult)]
    cfg, r = PcieConfig(bytes, iters, int(verify), workgroups, 0), PcieResult()
    rc = L.gmd_pcie_test(device, C.byref(cfg), C.byref(r))
    if rc:
        raise DiagError(rc, "gmd_pcie_test")
    return _dict(r)


def pcie_link_gbps(gen, width):
    """What one direction of a PCIe link of that generation and width carries at most, GB/s."""
    L = _lib()
    L.gmd_pcie_link_gbps.restype = C.c_double
    L.gmd_pcie_link_gbps.argtypes = [C.c_int, C.c_int]
    return L.gmd_pcie_link_gbps(gen, width)


def _expect(expect):
    e = LinkExpect()
    for k, v in (expect or {}).items():
        setattr(e, k, v)
    return e


def link_expected_gbps(mode="uni", peers=1, **expect):
    """Expected rate per direction of one link under a traffic pattern, times `peers` (see gmd_link_expect_t)."""
    L = _lib()
    L.gmd_link_expected_gbps.restype = C.c_double
    L.gmd_link_expected_gbps.argtypes = [C.POINTER(LinkExpect), C.c_int, C.c_int]
    return L.gmd_link_expected_gbps(C.byref(_expect(expect)), LINK_MODES[mode], peers)


def rccl_expected_busbw(op, ndev, **expect):
    """Expected peak bus bandwidth of a collective on ndev GPUs, from the link rate (0 for one GPU)."""
    L = _lib()
    L.gmd_rccl_expected_busbw.restype = C.c_double
    L.gmd_rccl_expected_busbw.argtypes = [C.POINTER(LinkExpect), C.c_int, C.c_int]
    return L.gmd_rccl_expected_busbw(C.byref(_expect(expect)), COLLECTIVES[op], ndev)


def grade_against(value, expected, warn_ratio=0.90, fail_ratio=0.75):
    L = _lib()
    L.gmd_grade_against.argtypes = [C.c_double, C.c_double, C.c_double, C.c_double]
    return L.gmd_grade_against(value, expected, warn_ratio, fail_ratio)


def outlier_grade(values, warn_ratio=0.0, fail_ratio=0.0):
    """Each value against the median of the others: (worst, [health], [median of the others])."""
    L = _lib()
    L.gmd_outlier_grade.argtypes = [C.c_int, C.POINTER(C.c_double), C.c_double, C.c_double, C.POINTER(C.c_int32),
                                    C.POINTER(C.c_double)]
    n = len(values)
    v = (C.c_double * n)(*[float(x) for x in values])
    h, m = (C.c_int32 * n)(), (C.c_double * n)()
    rc = L.gmd_outlier_grade(n, v, warn_ratio, fail_ratio, h, m)
    if rc < 0:
        raise DiagError(rc, "gmd_outlier_grade")
    return rc, list(h), list(m)


def link_diagnose(uni, bidi=None, all_=None, bad=None, **expect):
    """The link matrices of the three patterns read together (host arithmetic, no GPU): returns (worst_health,
    [finding dict]); a finding has kind (no_access, corrupt_data, slow_link, half_duplex, slow_under_load, low_egress,
    low_ingress), pattern, src, dst (-1 = the whole GPU on that side), health, gbps, expected_gbps, message."""
    L = _lib()
    L.gmd_link_diagnose.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                    C.POINTER(C.c_uint64), C.POINTER(LinkExpect), C.POINTER(LinkFinding),
                                    C.POINTER(C.c_int)]
    L.gmd_link_finding_name.restype = C.c_char_p
    n = len(uni)

    def flat(m, t):
        return (t * (n * n))(*[x for row in m for x in row]) if m is not None else None

    cap = 3 * n * n + 2 * n + 1
    out = (LinkFinding * cap)()
    cnt = C.c_int(cap)
    rc = L.gmd_link_diagnose(n, flat(uni, C.c_double), flat(bidi, C.c_double), flat(all_, C.c_double),
                             flat(bad, C.c_uint64), C.byref(_expect(expect)), out, C.byref(cnt))
    if rc < 0:
        raise DiagError(rc, "gmd_link_diagnose")
    names = {v: k for k, v in LINK_MODES.items()}
    res = []
    for i in range(cnt.value):
        d = _dict(out[i])
        d["kind"] = L.gmd_link_finding_name(d["kind"]).decode()
        d["pattern"] = names[d.pop("mode")]
        res.append(d)
    return rc, res


def link_grade(gbps, bad=None, floor_gbps=0.0, warn_ratio=0.0, fail_ratio=0.0):
    """Grades a link matrix (host arithmetic, no GPU): returns (worst_health, [verdict dict per ordered pair]).
    Health is 0 pass / 10 warn / 20 fail; a pair fails without peer access, with corrupt data, under the absolute
    floor or under fail_ratio (0.75) of the other pairs' median, and warns under warn_ratio (0.90) of it."""
    L = _lib()
    L.gmd_link_grade.argtypes = [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(LinkRule),
                                 C.POINTER(LinkVerdict), C.POINTER(C.c_int)]
    n = len(gbps)
    m = (C.c_double * (n * n))(*[float(x) for row in gbps for x in row])
    b = (C.c_uint64 * (n * n))(*[int(x) for row in bad for x in row]) if bad else None
    rule = LinkRule(floor_gbps, warn_ratio, fail_ratio)
    out = (LinkVerdict * max(1, n * (n - 1)))()
    cnt = C.c_int(n * (n - 1))
    rc = L.gmd_link_grade(n, m, b, C.byref(rule), out, C.byref(cnt))
    if rc < 0:
        raise DiagError(rc, "gmd_link_grade")
    return rc, [_dict(out[i]) for i in range(cnt.value)]


def gemm_launch(dtype, a_ptr, b_ptr, c_ptr, m, n, k, stream=None, scale_a=None, scale_b=None):
    """C[M,N] (bf16) = A[M,K] @ B[N,K]^T on caller memory.  dtype "mxfp8" takes the two packed E8M0 scale tensors
    (mx_pack_scales) as scale_a / scale_b device pointers; "fp8" is unscaled e4m3.  M and N in multiples of 8 (256
    for "mxfp8"), K in multiples of 64 ("bf16") or 128; operands 16-byte aligned."""
    if dtype == "mxfp8":
        if scale_a is None or scale_b is None:
            raise DiagError(-1, "gmd_gemm_launch_mx: mxfp8 needs scale_a and scale_b (packed E8M0 scales)")
        rc = _lib().gmd_gemm_launch_mx(C.c_void_p(a_ptr), C.c_void_p(b_ptr), C.c_void_p(scale_a), C.c_void_p(scale_b),
                                       C.c_void_p(c_ptr), m, n, k, C.c_void_p(stream))
    else:
        rc = _lib().gmd_gemm_launch(DTYPES[dtype], a_ptr, b_ptr, c_ptr, m, n, k, stream)
    if rc:
        raise DiagError(rc, f"gmd_gemm_launch({dtype})")


def mx_pack_scales(src_ptr, dst_ptr, rows, k, stream=None):
    """Row-major uint8 E8M0 scales [rows][k/32] -> the packed layout the MX-fp8 GEMM reads (same byte count);
    rows in multiples of 128, k of 128."""
    rc = _lib().gmd_mx_pack_scales(C.c_void_p(src_ptr), C.c_void_p(dst_ptr), rows, k, C.c_void_p(stream))
    if rc:
        raise DiagError(rc, "gmd_mx_pack_scales")
