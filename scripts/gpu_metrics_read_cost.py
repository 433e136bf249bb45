#!/usr/bin/env python3
"""What the kernel charges for one read of the bulk metrics table (sysfs gpu_metrics), next to the device library's
call that wraps it: tells how much of the sampling cost is the library's own."""
import glob
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    out = {}
    paths = sorted(glob.glob("/sys/class/drm/card*/device/gpu_metrics"))
    out["paths"] = paths
    if paths:
        p = paths[0]
        fd = os.open(p, os.O_RDONLY)
        raw = os.pread(fd, 8192, 0)
        out["bytes"] = len(raw)
        out["structure_size"] = int.from_bytes(raw[0:2], "little")
        out["format_revision"], out["content_revision"] = raw[2], raw[3]
        for name, fn in (("pread_kept_fd", lambda: os.pread(fd, 8192, 0)),
                         ("open_read_close", lambda: open(p, "rb").read())):
            n = 2000
            c0, t0 = time.process_time(), time.perf_counter()
            for _ in range(n):
                fn()
            out[name + "_us_wall"] = round((time.perf_counter() - t0) / n * 1e6, 2)
            out[name + "_us_cpu"] = round((time.process_time() - c0) / n * 1e6, 2)
        os.close(fd)
        out["raw_hex"] = raw.hex()
    from k8s_gpu_monitor_amd import smi
    smi.init()
    try:
        n = 2000
        c0, t0 = time.process_time(), time.perf_counter()
        for _ in range(n):
            smi.device_status(0)
        out["smi_device_status_us_wall"] = round((time.perf_counter() - t0) / n * 1e6, 2)
        out["smi_device_status_us_cpu"] = round((time.process_time() - c0) / n * 1e6, 2)
        out["status"] = {k: v for k, v in smi.device_status(0).items() if isinstance(v, (int, float, str, type(None)))}
    finally:
        smi.shutdown()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
