"""The exported surface of the reference's two Go packages, name by name, against this tree's `smi` and `hostengine`
packages: every function, method, type and struct member a program written against the reference can name has a
counterpart here, either under the same name or under the rename this file lists (and docs/migration.md explains).

The reference surface is held below as plain name lists (bindings/go/nvml/{nvml,bindings}.go and
bindings/go/dcgm/*.go, exported identifiers outside the test files).  When the reference tree is on the machine the
lists are regenerated from it and compared, so they cannot drift from what they claim to be."""
import glob
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GO = os.path.join(ROOT, "bindings", "go")
REFERENCE = "/root/reference/bindings/go"

NVML_FUNCS = ["DeleteEventSet", "Device.GetAllRunningProcesses", "Device.GetComputeRunningProcesses", "Device.GetDeviceMode",
              "Device.GetGraphicsRunningProcesses", "Device.Status", "GetCudaDriverVersion", "GetDeviceCount", "GetDriverVersion",
              "GetNVLink", "GetP2PLink", "Init", "ModeState.String", "NewDevice", "NewDeviceLite", "NewEventSet",
              "P2PLinkType.String", "PerfState.String", "ProcessType.String", "RegisterEvent", "RegisterEventForDevice", "Shutdown",
              "ThrottleReason.String", "WaitForEvent"]
NVML_FIELDS = {
    "Accounting": ["Mode", "BufferSize"], "ClockInfo": ["Cores", "Memory"], "CudaComputeCapabilityInfo": ["Major", "Minor"],
    "Device": ["UUID", "Path", "Model", "Power", "Memory", "CPUAffinity", "PCI", "Clocks", "Topology", "CudaComputeCapability"],
    "DeviceMemory": ["Used", "Free"], "DeviceMode": ["DisplayInfo", "Persistence", "AccountingInfo"],
    "DeviceStatus": ["Power", "Temperature", "Utilization", "Memory", "Clocks", "PCI", "Processes", "Throttle", "Performance"],
    "Display": ["Mode", "Active"], "ECCErrorsInfo": ["L1Cache", "L2Cache", "Device"], "Event": ["UUID", "Etype", "Edata"],
    "MemoryInfo": ["Global", "ECCErrors"], "P2PLink": ["BusID", "Link"], "PCIInfo": ["BusID", "BAR1", "Bandwidth"],
    "PCIStatusInfo": ["BAR1Used", "Throughput"], "PCIThroughputInfo": ["RX", "TX"],
    "ProcessInfo": ["PID", "Name", "MemoryUsed", "Type"], "UtilizationInfo": ["GPU", "Memory", "Encoder", "Decoder"]}
NVML_OTHER_TYPES = ["EventSet", "ModeState", "P2PLinkType", "PerfState", "ProcessType", "ThrottleReason"]

DCGM_FUNCS = ["GetAllDeviceCount", "GetDeviceInfo", "GetDeviceStatus", "GetDeviceTopology", "GetProcessInfo", "GetSupportedDevices",
              "HealthCheckByGpuId", "Init", "Introspect", "P2PLinkType.PCIPaths", "PerfState.String", "Policy", "Shutdown",
              "Time.String", "ViolationRegistration", "WatchPidFields"]
DCGM_FIELDS = {
    "ClockInfo": ["Cores", "Memory"], "DcgmStatus": ["Memory", "CPU"],
    "Device": ["GPU", "DCGMSupported", "UUID", "Power", "PCI", "Clocks", "Identifiers", "Topology", "CPUAffinity"],
    "DeviceHealth": ["GPU", "Status", "Watches"],
    "DeviceIdentifiers": ["Brand", "Model", "Serial", "Vbios", "InforomImageVersion", "DriverVersion"],
    "DeviceStatus": ["Power", "Temperature", "Utilization", "Memory", "Clocks", "PCI", "Performance", "FanSpeed"],
    "ECCErrorsInfo": ["SingleBit", "DoubleBit"], "MemoryInfo": ["GlobalUsed", "ECCErrors"], "P2PLink": ["GPU", "BusID", "Link"],
    "PCIInfo": ["BusID", "BAR1", "FBTotal", "Bandwidth"], "PCIStatusInfo": ["BAR1Used", "Throughput", "FBUsed"],
    "PCIThroughputInfo": ["Rx", "Tx", "Replays"], "PolicyViolation": ["Condition", "Timestamp", "Data"],
    "ProcessInfo": ["GPU", "PID", "Name", "ProcessUtilization", "PCI", "Memory", "GpuUtilization", "Clocks", "Violations",
                    "XIDErrors"],
    "ProcessUtilInfo": ["StartTime", "EndTime", "EnergyConsumed", "SmUtil", "MemUtil"],
    "SystemWatch": ["Type", "Status", "Error"], "UtilizationInfo": ["GPU", "Memory", "Encoder", "Decoder"],
    "ViolationTime": ["Power", "Thermal", "Reliability", "BoardLimit", "LowUtilization", "SyncBoost"],
    "XIDErrorInfo": ["NumErrors", "Timestamp"]}
DCGM_OTHER_TYPES = ["P2PLinkType", "PerfState", "Time"]

# Renames: where the NVIDIA word names a thing the AMD part does not have, the AMD word is used.  Everything not
# listed keeps the reference's name.  "Type.Member" entries rename a struct member, bare entries a function or type;
# None marks the one name with no counterpart by design.
SMI_RENAMES = {
    "GetCudaDriverVersion": "GetRuntimeVersion",              # the HIP runtime's version
    "GetNVLink": "GetXGMILink",
    "CudaComputeCapabilityInfo": "GfxVersionInfo",            # gfx950 = {9, 5, 0}
    "Device.CudaComputeCapability": "Device.GfxTarget",       # "gfx950"; (*Device).GfxVersion() gives the numbers
    "PCIInfo.BAR1": "PCIInfo.VisibleVRAM",                    # the CPU-visible aperture
    "PCIStatusInfo.BAR1Used": "PCIStatusInfo.VisibleVRAMUsed",
    "ECCErrorsInfo.L1Cache": "ECCErrorsInfo.Correctable",     # the device library counts by severity, not by cache level
    "ECCErrorsInfo.L2Cache": "ECCErrorsInfo.Uncorrectable",
    "ECCErrorsInfo.Device": "ECCErrorsInfo.Uncorrectable",
}
HOSTENGINE_RENAMES = {
    "ViolationRegistration": None,  # the cgo callback the reference had to export; here goViolationNotify, unexported use
    "Device.DCGMSupported": "Device.Supported",
    "DeviceIdentifiers.InforomImageVersion": "DeviceIdentifiers.Firmware",
    "PCIInfo.BAR1": "PCIInfo.VisibleVRAM",
    "PCIInfo.FBTotal": "PCIInfo.VRAMTotal",
    "PCIStatusInfo.BAR1Used": "PCIStatusInfo.VisibleVRAMUsed",
    "PCIStatusInfo.FBUsed": "MemoryInfo.GlobalUsed",          # frame buffer in use is HBM in use: one number, one place
    "ECCErrorsInfo.SingleBit": "ECCErrorsInfo.Correctable",
    "ECCErrorsInfo.DoubleBit": "ECCErrorsInfo.Uncorrectable",
    "ProcessUtilInfo.SmUtil": "ProcessUtilInfo.GfxUtil",
}


def go_surface(directory):
    """(functions and Type.Method names, type names incl. aliases, {struct: [exported members]}) of a Go package."""
    funcs, types, fields = set(), set(), {}
    for path in glob.glob(os.path.join(directory, "*.go")):
        if path.endswith("_test.go"):
            continue
        src = open(path).read()
        for m in re.finditer(r"^func (?:\(\w+ \*?(\w+)\) )?([A-Z]\w*)\(", src, re.M):
            funcs.add((m.group(1) + "." if m.group(1) else "") + m.group(2))
        types |= set(re.findall(r"^type ([A-Z]\w*)", src, re.M))
        for block in re.findall(r"^type \($(.*?)^\)", src, re.M | re.S):  # grouped declarations (the alias block)
            types |= set(re.findall(r"^\s+([A-Z]\w*)\s+=?\s*\w", block, re.M))
        for m in re.finditer(r"^type ([A-Z]\w*) struct \{\n(.*?)^\}", src, re.M | re.S):
            members = []
            for line in m.group(2).splitlines():
                parts = line.split("//")[0].split()
                if len(parts) >= 2:
                    members += [n.strip(",") for n in parts[:-1] if n[0].isupper()]
            fields[m.group(1)] = members
    return funcs, types, fields


def missing(ref_funcs, ref_fields, ref_types, renames, ours):
    funcs, types, fields = ours
    out = []
    for f in ref_funcs:
        want = renames.get(f, f)
        if want is not None and want not in funcs:
            out.append(f"function {f} -> {want}")
    for t in list(ref_fields) + ref_types:
        want = renames.get(t, t)
        if want not in types:
            out.append(f"type {t} -> {want}")
    for t, members in ref_fields.items():
        for member in members:
            want = renames.get(f"{t}.{member}", f"{renames.get(t, t)}.{member}")
            wt, wm = want.split(".")
            wt = {"DcgmStatus": "Status", "XIDErrorInfo": "CriticalEvents"}.get(wt, wt)  # aliases resolve to their struct
            if wm not in fields.get(wt, []):
                out.append(f"member {t}.{member} -> {want}")
    return out


def test_smi_package_covers_the_reference_nvml_surface():
    assert missing(NVML_FUNCS, NVML_FIELDS, NVML_OTHER_TYPES, SMI_RENAMES, go_surface(os.path.join(GO, "smi"))) == []


def test_hostengine_package_covers_the_reference_dcgm_surface():
    assert missing(DCGM_FUNCS, DCGM_FIELDS, DCGM_OTHER_TYPES, HOSTENGINE_RENAMES, go_surface(os.path.join(GO, "hostengine"))) == []


def test_a_dropped_member_or_function_is_reported():
    funcs, types, fields = go_surface(os.path.join(GO, "smi"))
    fields = dict(fields, DeviceStatus=[m for m in fields["DeviceStatus"] if m != "Throttle"])
    got = missing(NVML_FUNCS, NVML_FIELDS, NVML_OTHER_TYPES, SMI_RENAMES, (funcs - {"GetXGMILink"}, types - {"PerfState"}, fields))
    assert got == ["function GetNVLink -> GetXGMILink", "type PerfState -> PerfState", "member DeviceStatus.Throttle -> DeviceStatus.Throttle"]


def test_every_rename_is_explained_in_the_migration_guide():
    guide = open(os.path.join(ROOT, "docs", "migration.md")).read()
    for table in (SMI_RENAMES, HOSTENGINE_RENAMES):
        for old, new in table.items():
            assert old.split(".")[-1] in guide, old
            if new:
                assert new.split(".")[-1] in guide, new


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="reference tree not on this machine")
def test_the_name_lists_are_the_reference_surface():
    for pkg, funcs, fields, other in (("nvml", NVML_FUNCS, NVML_FIELDS, NVML_OTHER_TYPES), ("dcgm", DCGM_FUNCS, DCGM_FIELDS, DCGM_OTHER_TYPES)):
        f, t, fl = go_surface(os.path.join(REFERENCE, pkg))
        assert sorted(f) == sorted(funcs), pkg
        assert fl == fields, pkg
        assert sorted(t) == sorted(list(fields) + other), pkg
