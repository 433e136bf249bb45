"""What the host's device library answers for the optional entry points (per-link xGMI metrics and status, xGMI
error status, PCIe throughput, violation status, CPER entries) and how long each call takes.
Writes gpurun_out/amdsmi_optional.json."""
import json
import os
import time

import amdsmi

out = {}
amdsmi.amdsmi_init()
h = amdsmi.amdsmi_get_processor_handles()[0]


def probe(name, fn, *a, **kw):
    t0 = time.time()
    try:
        r = fn(h, *a, **kw)
        out[name] = {"ok": True, "seconds": round(time.time() - t0, 4), "result": r}
    except Exception as e:  # noqa: BLE001
        out[name] = {"ok": False, "seconds": round(time.time() - t0, 4), "error": repr(e)[:300]}
    print(name, json.dumps(out[name], default=str)[:600], flush=True)


probe("link_metrics", amdsmi.amdsmi_get_link_metrics)
probe("xgmi_link_status", amdsmi.amdsmi_get_gpu_xgmi_link_status)
probe("xgmi_error_status", amdsmi.amdsmi_gpu_xgmi_error_status)
probe("xgmi_info", amdsmi.amdsmi_get_xgmi_info)
probe("pci_throughput", amdsmi.amdsmi_get_gpu_pci_throughput)
probe("violation_status", amdsmi.amdsmi_get_violation_status)
probe("violation_status_again", amdsmi.amdsmi_get_violation_status)
try:
    t0 = time.time()
    r = amdsmi.amdsmi_get_gpu_cper_entries(h, 0x7, 1 << 16, 0)
    out["cper"] = {"ok": True, "seconds": round(time.time() - t0, 4), "result": str(r)[:500]}
except Exception as e:  # noqa: BLE001
    out["cper"] = {"ok": False, "error": repr(e)[:300]}
print("cper", out["cper"], flush=True)
amdsmi.amdsmi_shut_down()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/amdsmi_optional.json", "w"), indent=1, default=str)
