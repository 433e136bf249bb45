"""What the kernel says about this process's use of the GPU, without the KFD process table.

Opens the device through PyTorch, puts some memory and some work on it, then prints every
/proc/self/fdinfo entry of an amdgpu or kfd file descriptor, the readability of the KFD process
table, and the sysfs nodes that name the card behind a render node.  Output: gpurun_out/fdinfo_probe.txt
"""
import glob
import os
import sys
import time

import torch

out = open(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/fdinfo_probe.txt", "w")


def p(*a):
    print(*a, file=out)
    out.flush()


x = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
a = torch.randn(4096, 4096, device="cuda", dtype=torch.bfloat16)
for _ in range(200):
    b = a @ a
torch.cuda.synchronize()
p("pid", os.getpid(), "uid", os.getuid())
for fd in sorted(os.listdir("/proc/self/fd"), key=int):
    try:
        tgt = os.readlink(f"/proc/self/fd/{fd}")
    except OSError:
        continue
    if "dri" in tgt or "kfd" in tgt:
        p(f"--- fd {fd} -> {tgt}")
        try:
            p(open(f"/proc/self/fdinfo/{fd}").read())
        except OSError as e:
            p("  unreadable:", e)
t0 = time.time()
for _ in range(2000):
    b = a @ a
torch.cuda.synchronize()
p("--- after", time.time() - t0, "s more work")
for fd in sorted(os.listdir("/proc/self/fd"), key=int):
    try:
        tgt = os.readlink(f"/proc/self/fd/{fd}")
    except OSError:
        continue
    if "renderD" in tgt:
        p(f"--- fd {fd} -> {tgt}")
        p(open(f"/proc/self/fdinfo/{fd}").read())
p("--- kfd proc table")
for path in ["/sys/class/kfd/kfd/proc", f"/sys/class/kfd/kfd/proc/{os.getpid()}"]:
    try:
        p(path, os.listdir(path)[:20])
    except OSError as e:
        p(path, "->", e)
p("--- other processes' fdinfo")
n_ok = n_deny = 0
for d in glob.glob("/proc/[0-9]*"):
    try:
        os.listdir(d + "/fdinfo")
        n_ok += 1
    except OSError:
        n_deny += 1
p("readable", n_ok, "denied", n_deny)
p("--- render nodes")
for r in sorted(glob.glob("/sys/class/drm/renderD*")):
    try:
        p(r, "->", os.path.realpath(r + "/device"))
    except OSError as e:
        p(r, e)
p("--- /dev/dri", os.listdir("/dev/dri") if os.path.isdir("/dev/dri") else None)
