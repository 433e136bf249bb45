from .device import (ClockInfo, Device, DeviceStatus, MemoryInfo, P2PLink, PCIInfo, ProcessInfo,  # noqa: F401
                     UtilizationInfo, device_status, new_device)
