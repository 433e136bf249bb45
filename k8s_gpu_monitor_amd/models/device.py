"""Device data model with the shape of the reference's ``nvml`` Go package
(bindings/go/nvml/nvml.go:23-266, 274-604): ``Device`` (static identity),
``DeviceStatus`` (one telemetry read), process lists, link classification.
Optional values are ``None`` instead of nil pointers; units follow the
reference (W, MiB, MHz, MB/s) with AMD semantics where NVIDIA's do not exist
(``path`` is the DRM render node, links are xGMI, the compute capability is
the gfx target).
"""
from dataclasses import dataclass, field
from typing import List, Optional

from .. import _abi as A
from .. import engine as E


@dataclass
class ClockInfo:
    cores: Optional[int] = None   # MHz
    memory: Optional[int] = None  # MHz


@dataclass
class PCIInfo:
    bus_id: str = ""
    visible_vram: Optional[int] = None  # MiB, BAR1 analog
    bandwidth: Optional[int] = None     # MB/s


@dataclass
class P2PLink:
    bus_id: str
    link: str  # "XGMI1", "PIX", "SYS", ...
    description: str


@dataclass
class Device:
    index: int
    uuid: str
    path: str
    model: Optional[str]
    power: Optional[int]   # W, enforced cap
    memory: Optional[int]  # MiB
    cpu_affinity: Optional[int]  # NUMA node
    pci: PCIInfo
    clocks: ClockInfo
    gfx_target: Optional[str]
    num_xcd: int = 0
    num_cu: int = 0
    compute_partition: str = ""
    memory_partition: str = ""
    topology: List[P2PLink] = field(default_factory=list)

    api: object = field(default=E, repr=False, compare=False)  # engine (cached) or smi (direct) module

    def status(self) -> "DeviceStatus":
        return device_status(self.index, api=self.api)

    def processes(self) -> List["ProcessInfo"]:
        """Empty where the process table cannot be read by this user (reported as not supported)."""
        from .._lib import GpumonError
        from .. import abi as A
        try:
            procs = self.api.device_processes(self.index)
        except GpumonError as e:
            if e.status in (A.GM_ST_NOT_SUPPORTED, A.GM_ST_NO_PERMISSION):
                return []
            raise
        return [ProcessInfo(p["pid"], p["name"], _mib(p["vram_bytes"]), "C") for p in procs]

    @property
    def gfx_version(self):
        """(major, minor, stepping) of the gfx target — gfx950 -> (9, 5, 0), gfx90a -> (9, 0, 10) — the
        counterpart of the reference's CudaComputeCapability (bindings/go/nvml/nvml.go:196-199); None
        when the target is unknown."""
        t = (self.gfx_target or "").split(":")[0]
        if not t.startswith("gfx") or len(t) < 6:
            return None
        try:
            return int(t[3:-2]), int(t[-2], 16), int(t[-1], 16)
        except ValueError:
            return None


@dataclass
class UtilizationInfo:
    gpu: Optional[int] = None
    memory: Optional[int] = None
    encoder: Optional[int] = None  # always None on MI355X: no encoder
    decoder: Optional[int] = None


@dataclass
class MemoryInfo:
    used: Optional[int] = None  # MiB
    free: Optional[int] = None  # MiB
    ecc_correctable: Optional[int] = None
    ecc_uncorrectable: Optional[int] = None


@dataclass
class ProcessInfo:
    pid: int
    name: str
    memory_used: int  # MiB
    type: str


@dataclass
class DeviceStatus:
    power: Optional[int]        # W
    temperature: Optional[int]  # °C hotspot
    utilization: UtilizationInfo
    memory: MemoryInfo
    clocks: ClockInfo
    pcie_rx: Optional[int]      # MB
    pcie_tx: Optional[int]
    throttle: Optional[int]     # bitset, A.GM_THROTTLE_*
    performance: Optional[int]
    processes: List[ProcessInfo] = field(default_factory=list)

    THROTTLE_NAMES = {1: "Power cap", 2: "Thermal", 4: "HBM thermal", 8: "VR thermal", 16: "PROCHOT",
                      32: "Low utilization", 64: "Host limit"}

    PERF_LEVEL_NAMES = ("auto", "low", "high", "manual", "stable_std", "stable_peak", "stable_min_mclk", "stable_min_sclk",
                        "determinism")

    def performance_name(self) -> str:
        """The driver's performance level by name: the counterpart of PerfState.String() (bindings/go/nvml/nvml.go:106-111),
        which prints P0..P15; the amdgpu driver has named levels."""
        if self.performance is None or not 0 <= self.performance < len(self.PERF_LEVEL_NAMES):
            return "Unknown"
        return self.PERF_LEVEL_NAMES[self.performance]

    def throttle_reasons(self) -> List[str]:
        if not self.throttle:
            return ["No clocks throttling"] if self.throttle == 0 else []
        return [n for b, n in self.THROTTLE_NAMES.items() if self.throttle & b]


def new_device(index: int, lite: bool = False, api=E) -> Device:
    """api: :mod:`engine` (values from the sampler cache) or :mod:`smi` (direct device reads)."""
    a = api.device_attributes(index)
    if not a["supported"] or not a["uuid"]:
        raise E.GpumonError(A.GM_ST_GPU_NOT_SUPPORTED, f"device {index}")
    nanv = lambda x: None if x != x else int(round(x))  # noqa: E731
    opt = lambda x: None if x is None or x < 0 else x   # noqa: E731
    dev = Device(
        index=index, uuid=a["uuid"], path=a["device_path"], model=a["name"] or None,
        power=None if lite else nanv(a["power_cap_w"]), memory=None if lite else opt(a["vram_total_mib"]),
        cpu_affinity=opt(a["numa_node"]),
        pci=PCIInfo(a["pci_bdf"], None if lite else opt(a["vis_vram_total_mib"]),
                    None if lite else opt(a["pcie_bandwidth_mbs"])),
        clocks=ClockInfo(None if lite else opt(a["gfx_clock_max_mhz"]), None if lite else opt(a["mem_clock_max_mhz"])),
        gfx_target=a["gfx_target"] or None, num_xcd=a["num_xcd"], num_cu=a["num_cu"],
        compute_partition=a["compute_partition"], memory_partition=a["memory_partition"], api=api)
    if not lite:
        from .._lib import lib
        for p in api.device_topology(index)["peers"]:
            dev.topology.append(P2PLink(p["pci_bdf"], p["link_name"], lib().gm_link_type_description(p["link"]).decode()))
    return dev


A_PROC_MEM_UNKNOWN = 0xFFFFFFFFFFFFFFFF


def _mib(vram_bytes):
    """MiB, or None for a process that is only known to hold the device open (GM_PROC_MEM_UNKNOWN)."""
    return None if vram_bytes == A_PROC_MEM_UNKNOWN else vram_bytes >> 20


def device_status(index: int, api=E) -> DeviceStatus:
    s = api.device_status(index)
    kb_to_mb = lambda x: None if x is None else x // 1000  # noqa: E731
    procs = [ProcessInfo(p["pid"], p["name"], _mib(p["vram_bytes"]), "C") for p in api.device_processes(index)]
    return DeviceStatus(
        power=None if s["power_w"] is None else int(s["power_w"]), temperature=s["temperature_c"],
        utilization=UtilizationInfo(s["gfx_util"], s["mem_util"], s["enc_util"], s["dec_util"]),
        memory=MemoryInfo(s["vram_used_mib"], s["vram_free_mib"], s["ecc_ce"], s["ecc_ue"]),
        clocks=ClockInfo(s["gfx_clock_mhz"], s["mem_clock_mhz"]),
        pcie_rx=kb_to_mb(s["pcie_rx_kb"]), pcie_tx=kb_to_mb(s["pcie_tx_kb"]),
        throttle=s["throttle_status"], performance=s["perf_level"], processes=procs)
