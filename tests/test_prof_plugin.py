"""The optional counter plugin (libgpumon_prof.so) where there is no GPU: it must build, refuse
politely, and never get in the way of an exporter that was asked to use it."""
import os
import re
import signal
import subprocess
import time
import urllib.request

from k8s_gpu_monitor_amd._lib import BUILD_DIR, binary
from k8s_gpu_monitor_amd.utils import promparse


def test_plugin_and_probe_are_built():
    assert os.path.exists(os.path.join(BUILD_DIR, "libgpumon_prof.so"))
    assert os.path.exists(binary("gpumon-prof-probe"))


def test_probe_without_a_device_fails_with_a_message_not_a_hang():
    import torch
    if torch.cuda.is_available():  # on a GPU box the probe works; the GPU tests cover it
        return
    r = subprocess.run([binary("gpumon-prof-probe"), "--count", "1"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1
    assert "gmp_open:" in r.stderr


def test_probe_usage():
    r = subprocess.run([binary("gpumon-prof-probe"), "--bogus"], capture_output=True, text=True, timeout=20)
    assert r.returncode == 2 and "usage:" in r.stderr


def test_exporter_flag_is_harmless_on_a_backend_without_counters():
    """--prof-counters implies -p; the scripted backend has its own profiling values and never loads the plugin."""
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2,seed=3")
    p = subprocess.Popen([binary("gpumon-exporter"), "-e", "--prof-counters", "-o", "", "-d", "200", "--listen", "127.0.0.1:0"],
                         stderr=subprocess.PIPE, text=True, env=env)
    try:
        port = None
        for _ in range(100):
            line = p.stderr.readline()
            m = re.search(r"Serving /metrics on port (\d+)", line or "")
            if m:
                port = int(m.group(1))
                break
        assert port
        time.sleep(0.5)
        fam = promparse.parse(urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=5).read().decode())
        assert len(fam["dcgm_fi_prof_pipe_tensor_active"].samples) == 2
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=15)


def test_python_wrapper_refuses_politely_without_a_device():
    """k8s_gpu_monitor_amd.ops.prof in a process of its own (the plugin opens the compute runtime)."""
    import sys
    import torch
    if torch.cuda.is_available():
        return
    code = ("from k8s_gpu_monitor_amd.ops import prof\n"
            "try:\n    prof.open()\nexcept prof.ProfError as e:\n    print('refused:', e)\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60,
                       cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0 and "refused:" in r.stdout, (r.stdout, r.stderr)
