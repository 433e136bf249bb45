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


# --------------------------------------------------- sampled mode: probe child --
def _exporter_with_probe(tmp_path, script_body, period="1"):
    probe = tmp_path / "probe.sh"
    probe.write_text("#!/bin/sh\n" + script_body)
    probe.chmod(0o755)
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2,seed=3,steady=1", GPUMON_PROF_PROBE=str(probe),
               GPUMON_PROF_PERIOD_S=period)
    p = subprocess.Popen([binary("gpumon-exporter"), "-e", "--prof-counters=child", "-o", "", "-d", "200", "--listen", "127.0.0.1:0"],
                         stderr=subprocess.PIPE, text=True, env=env)
    port = None
    for _ in range(100):
        line = p.stderr.readline()
        m = re.search(r"Serving /metrics on port (\d+)", line or "")
        if m:
            port = int(m.group(1))
            break
    assert port
    return p, port


def _scrape(port):
    return promparse.parse(urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=5).read().decode())


def test_child_mode_exports_what_the_probe_measured_and_blanks_the_rest(tmp_path):
    """--prof-counters=child: the daemon starts `probe --once --interval-ms N` every period and exports the line of
    each GPU under that GPU (matched by PCI address); a GPU the probe did not report and a value it printed as nan
    are absent from the scrape, never 0."""
    log = tmp_path / "calls.log"
    body = (f'echo "$@" >> {log}\n'
            'echo "agents: 1"\n'
            'echo "sample pci=0000:05:00.0 window_s=0.750 gpu_active=99.1 cu_active=77.25 mem_pipe_active=nan mfma_active=42.5 clock_mhz=1934"\n')
    p, port = _exporter_with_probe(tmp_path, body)
    try:
        deadline = time.time() + 10
        fam = {}
        while time.time() < deadline:
            fam = _scrape(port)
            if "dcgm_fi_prof_pipe_tensor_active" in fam and fam["dcgm_fi_prof_pipe_tensor_active"].samples:
                break
            time.sleep(0.3)
        tensor = {l["gpu"]: v for l, v in fam["dcgm_fi_prof_pipe_tensor_active"].samples}
        sm = {l["gpu"]: v for l, v in fam["dcgm_fi_prof_sm_active"].samples}
        assert tensor == {"0": 42.5} and sm == {"0": 77.25}, (tensor, sm)
        # the values are duty-cycle samples, and the HELP text of exactly the counter-fed families says so
        assert fam["dcgm_fi_prof_pipe_tensor_active"].help.endswith("Sampled: about 0.75 s of counting every 1 s.")
        assert fam["dcgm_fi_prof_sm_active"].help.endswith("every 1 s.")
        assert "Sampled" not in fam["dcgm_fi_prof_gr_engine_active"].help
        assert "gpumon_prof_mem_pipe_active" not in fam or not fam["gpumon_prof_mem_pipe_active"].samples
        assert len(fam["dcgm_fi_prof_gr_engine_active"].samples) == 2  # not counter-derived on this backend
        time.sleep(2.5)
        calls = log.read_text().splitlines()
        assert len(calls) >= 2 and all(c.startswith("--once --interval-ms ") for c in calls), calls
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=15)


def test_child_mode_survives_a_failing_and_a_hanging_probe(tmp_path):
    """A probe that fails gives blanks, not a crash and not stale numbers; a probe that never ends is killed with
    the daemon (the exact child, by pid)."""
    p, port = _exporter_with_probe(tmp_path, 'echo "gmp_open: no device" >&2\nexit 1\n')
    try:
        time.sleep(2.0)
        fam = _scrape(port)
        assert "dcgm_fi_prof_pipe_tensor_active" not in fam or not fam["dcgm_fi_prof_pipe_tensor_active"].samples
        assert len(fam["dcgm_gpu_temp"].samples) == 2
    finally:
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=15) == 0
    pidfile = tmp_path / "probe.pid"
    p, port = _exporter_with_probe(tmp_path, f'echo $$ > {pidfile}\nexec sleep 300\n')
    try:
        for _ in range(50):
            if pidfile.exists() and pidfile.read_text().strip():
                break
            time.sleep(0.1)
        child = int(pidfile.read_text())
        assert len(_scrape(port)["dcgm_gpu_temp"].samples) == 2  # the sampler thread waits, the exporter does not
    finally:
        t0 = time.time()
        p.send_signal(signal.SIGTERM)
        assert p.wait(timeout=15) == 0 and time.time() - t0 < 5
    time.sleep(0.2)
    assert not os.path.exists(f"/proc/{child}") or open(f"/proc/{child}/stat").read().split()[2] == "Z"
