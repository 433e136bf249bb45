"""Captures `amd-smi` JSON (static / metric / list) next to our own direct-layer read of the
same GPU, for building oracle tests.  Writes gpurun_out/amdsmi_dump.json."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd import smi  # noqa: E402
from k8s_gpu_monitor_amd.utils import amdsmi_cli  # noqa: E402

out = {}
for section in ("list", "static", "metric"):
    r = subprocess.run([amdsmi_cli.binary(), section, "--json"], capture_output=True, text=True, timeout=120)
    out[section] = {"rc": r.returncode, "stdout": r.stdout, "stderr": r.stderr[-500:]}
r = subprocess.run([amdsmi_cli.binary(), "version"], capture_output=True, text=True, timeout=60)
out["version"] = r.stdout
smi.init()
out["ours"] = {"count": smi.device_count(), "driver": smi.driver_version(), "attributes": smi.device_attributes(0),
               "status": smi.device_status(0), "mode": smi.device_mode(0)}
smi.shutdown()
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/amdsmi_dump.json", "w"), indent=1, default=str)
print("ok")
