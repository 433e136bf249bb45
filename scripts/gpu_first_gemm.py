#!/usr/bin/env python3
"""Where does the first GEMM of a bench step lose its time?  Eight bf16 8192^3 launches, each between two HIP
events, after four different predecessors: an idle device, a 4 GiB stream copy with no host wait, the memtest
fill + verify with its host read-back (what bench.py does), and a 4 GiB copy followed by a host wait.
Prints one JSON object: microseconds per launch position, median over the repetitions."""
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from k8s_gpu_monitor_amd.ops import diag  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
st = torch.cuda.current_stream(dev)
sp = st.cuda_stream
D, NB, REPS = 8192, 4 << 30, 12
g = torch.Generator(device=dev).manual_seed(1)
A = torch.empty(D, D, dtype=torch.bfloat16, device=dev).normal_(generator=g)
B = torch.empty(D, D, dtype=torch.bfloat16, device=dev).normal_(generator=g)
Cm = torch.empty(D, D, dtype=torch.bfloat16, device=dev)
ha = torch.empty(NB // 4, dtype=torch.float32, device=dev).normal_(generator=g)
hc = torch.empty(NB // 4, dtype=torch.float32, device=dev)


def gemms():
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(9)]
    ev[0].record(st)
    for i in range(8):
        diag.gemm_launch("bf16", A.data_ptr(), B.data_ptr(), Cm.data_ptr(), D, D, D, sp)
        ev[i + 1].record(st)
    torch.cuda.synchronize()
    return [ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(8)]


def copy():
    diag.hbm_stream_launch("copy", ha.data_ptr(), 0, hc.data_ptr(), 0.0, NB // 4, "auto", sp)


def pre_idle():
    torch.cuda.synchronize()
    time.sleep(0.005)


def pre_copy_nowait():
    copy()


def pre_memtest_readback():
    diag.memtest_fill(hc.data_ptr(), NB // 8, "random", seed=3, stream=sp)
    diag.memtest_verify(hc.data_ptr(), NB // 8, "random", seed=3, stream=sp)


def pre_copy_wait():
    copy()
    torch.cuda.synchronize()


def pre_gemm_nowait():
    diag.gemm_launch("bf16", A.data_ptr(), B.data_ptr(), Cm.data_ptr(), D, D, D, sp)


out = {}
for name, pre in (("idle_5ms", pre_idle), ("copy_4GiB_no_host_wait", pre_copy_nowait), ("memtest_with_readback", pre_memtest_readback),
                  ("copy_4GiB_then_host_wait", pre_copy_wait), ("gemm_no_host_wait", pre_gemm_nowait)):
    rows = []
    for r in range(REPS + 2):
        pre()
        t = gemms()
        if r >= 2:
            rows.append(t)
    out[name] = [round(statistics.median(x[i] for x in rows), 1) for i in range(8)]
    print(name, out[name], flush=True)
print(json.dumps(out))
