"""Which VRAM-usage counters move on this box when 8 GiB are allocated?  (diagnostic script)"""
import glob
import os
import subprocess

import torch


def snap():
    out = {}
    for p in glob.glob("/sys/class/drm/card*/device/mem_info_*") + glob.glob("/sys/class/kfd/kfd/topology/nodes/*/mem_banks/*/used_memory"):
        try:
            out[p] = open(p).read().strip()
        except OSError as e:
            out[p] = f"ERR {e.errno}"
    free, total = torch.cuda.mem_get_info(0)
    out["hip_used"] = str(total - free)
    return out


torch.cuda.init()
a = snap()
x = torch.empty(8 << 30, dtype=torch.uint8, device="cuda").fill_(1)
torch.cuda.synchronize()
b = snap()
for k in sorted(a):
    print(f"{k}: {a[k]} -> {b[k]}")
for cmd in (["amd-smi", "metric", "-g", "0", "--mem-usage", "--json"], ["amd-smi", "static", "-g", "0", "--asic", "--json"],
            ["rocm-smi", "--showmeminfo", "vram", "--json"]):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=60)
        print("$", " ".join(cmd), "->", r.returncode)
        print(r.stdout[-1500:], r.stderr[-500:])
    except Exception as e:  # noqa: BLE001
        print("$", " ".join(cmd), "failed:", e)
print("uid", os.getuid())
