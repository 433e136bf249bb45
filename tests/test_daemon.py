"""Standalone host engine (gpumon-hostengine) + remote clients, and the CLIs.

Covers the three run modes of the reference's Init (bindings/go/dcgm/admin.go:26-30,
81-194): embedded, standalone (connect to a running daemon over a unix socket or
TCP) and start-hostengine (spawn a private daemon as a child)."""
import json
import os
import re
import signal
import socket
import subprocess
import time
import urllib.error
import urllib.request

import pytest

from conftest import bin_path
from k8s_gpu_monitor_amd import abi as A
from k8s_gpu_monitor_amd import engine as E

P = A.POLICY
ENV = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=4,seed=3")


def _wait_for(path_or_port, timeout=10):
    deadline = time.time() + timeout
    while time.time() < deadline:
        if isinstance(path_or_port, str):
            if os.path.exists(path_or_port):
                return
        else:
            try:
                socket.create_connection(("127.0.0.1", path_or_port), timeout=0.2).close()
                return
            except OSError:
                pass
        time.sleep(0.05)
    raise TimeoutError(path_or_port)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture
def daemon(tmp_path):
    sock = str(tmp_path / "he.sock")
    http = _free_port()
    pidf = str(tmp_path / "he.pid")
    p = subprocess.Popen([bin_path("gpumon-hostengine"), "--domain-socket", sock, "--http", f"127.0.0.1:{http}",
                          "--pid-file", pidf], env=ENV, stderr=subprocess.PIPE)
    _wait_for(sock)
    yield {"sock": sock, "http": http, "proc": p, "pid_file": pidf}
    if p.poll() is None:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)


def test_standalone_client_full_api(daemon):
    E.init(mode=E.STANDALONE, address=daemon["sock"], is_unix_socket=True)
    try:
        assert E.device_count() == 4
        assert E.device_attributes(1)["uuid"] == "GPU-00000003-fa4e-0355-0000-000000000001"
        assert len(E.device_topology(0)["peers"]) == 3
        g = E.group_create("remote", [0, 2])
        fg = E.field_group_create("remote", [A.FI["GPU_TEMP"], A.FI["POWER_USAGE"], A.FI["DEV_NAME"]])
        E.watch_fields(g, fg, update_freq_us=50_000)
        E.update_all_fields(True)
        v = E.latest_values(2, [A.FI["GPU_TEMP"], A.FI["POWER_USAGE"], A.FI["DEV_NAME"]])
        assert v[A.FI["GPU_TEMP"]] > 30 and v[A.FI["DEV_NAME"]] == "AMD Instinct MI355X"
        time.sleep(0.3)
        assert len(E.values_since(0, A.FI["GPU_TEMP"])) >= 3
        assert E.device_status(0)["temperature_c"] > 30
        E.health_set(g)
        E.fake_ctl("link_down gpu=2 link=1")
        E.update_all_fields(True)
        h = {r["gpu"]: r for r in E.health_check(g)}
        assert h[0]["status"] == "Healthy" and h[2]["status"] == "Failure"
        # asynchronous violations travel back over the same connection
        E.policy_set(g, P["THERMAL"], thermal_c=90)
        with E.PolicyRegistration(g, P["THERMAL"] | P["GPU_EVENT"]) as reg:
            E.fake_ctl("set gpu=0 field=150 value=96")
            E.update_all_fields(True)
            v = reg.wait(3000)
            assert v and v["name"] == "Thermal Limit" and v["data"]["temperature_c"] == 96
            E.fake_ctl("event gpu=2 code=5 msg=ras")
            v = reg.wait(3000)
            assert v and v["gpu"] == 2 and v["data"]["code"] == 5
        es = E.EventSet()
        E.fake_ctl("event gpu=1 code=4")
        ev = es.wait(3000)
        assert ev and (ev["gpu"], ev["code"]) == (1, 4)
        es.close()
        E.watch_pid_fields(update_freq_us=50_000)
        E.fake_ctl("proc_add gpu=3 pid=99 name=job")
        time.sleep(0.3)
        assert E.pid_info(99)[0]["gpu"] == 3
        # what the daemon's device layer says about its own sources travels too
        assert E.process_source(3) == "fake" and E.metrics_table_source(3) == "n/a"
        with pytest.raises(E.GpumonError):
            E.metrics_table_source(64)
        i = E.introspect()
        assert i["memory_kb"] > 1000  # the DAEMON's memory, not ours
        with E.Exporter(interval_ms=200) as ex:  # an exporter can run against a remote engine too
            ex.sweep()
            assert 'dcgm_gpu_temp{gpu="3"' in ex.render()
    finally:
        E.shutdown()


def test_daemon_cleans_up_after_a_client(daemon):
    E.init(mode=E.STANDALONE, address=daemon["sock"], is_unix_socket=True)
    g = E.group_create("leak-check", [0])
    E.shutdown()  # disconnect without destroying the group
    time.sleep(0.2)
    E.init(mode=E.STANDALONE, address=daemon["sock"], is_unix_socket=True)
    try:
        E.group_create("leak-check", [0])  # the name is free again: the daemon reaped the old group
        with pytest.raises(E.GpumonError):
            E.group_devices(g) if g != 1 else (_ for _ in ()).throw(E.GpumonError(A.GM_ST_NOT_FOUND))
    finally:
        E.shutdown()


def test_tcp_listener_and_rest_api(tmp_path):
    port, http = _free_port(), _free_port()
    p = subprocess.Popen([bin_path("gpumon-hostengine"), "-b", "127.0.0.1", "--port", str(port), "--http",
                          f"127.0.0.1:{http}", "--pid-file", str(tmp_path / "p.pid")], env=ENV, stderr=subprocess.PIPE)
    try:
        _wait_for(port)
        _wait_for(http)
        E.init(mode=E.STANDALONE, address=f"127.0.0.1:{port}")
        assert E.device_count() == 4
        E.shutdown()
        base = f"http://127.0.0.1:{http}"

        def get(path):
            return urllib.request.urlopen(base + path, timeout=5).read().decode()

        def code(path):
            try:
                urllib.request.urlopen(base + path, timeout=5)
                return 200
            except urllib.error.HTTPError as e:
                return e.code

        # routes of the reference's sample server (restApi/server.go:40-71), both prefixes
        txt = get("/dcgm/device/info/id/1")
        assert "UUID                   : GPU-00000003-fa4e-0355-0000-000000000001" in txt
        assert "GPU0 - (BusID)0000:05:00.0 - XGMI1" in txt
        info = json.loads(get("/dcgm/device/info/id/1/json"))
        assert info["GPU"] == 1 and info["PCI"]["FBTotal"] == 294896 and info["Identifiers"]["Model"] == "AMD Instinct MI355X"
        uuid = info["UUID"]
        assert json.loads(get(f"/dcgm/device/info/uuid/{uuid}/json"))["GPU"] == 1
        st = json.loads(get(f"/gpumon/device/status/uuid/{uuid}/json"))
        assert st["Temperature"] > 30 and st["Utilization"]["Encoder"] is None  # blank -> null, never a sentinel
        assert "Temperature (°C)        :" in get("/dcgm/device/status/id/0")
        assert json.loads(get("/dcgm/health/id/2/json")) == {"GPU": 2, "Status": "Healthy", "Watches": []}
        assert "Status             : Healthy" in get(f"/dcgm/health/uuid/{uuid}")
        hs = json.loads(get("/dcgm/status/json"))
        assert hs["Memory"] > 1000 and hs["CPU"] >= 0
        assert get("/dcgm/status").startswith("Memory(KB)      :")
        assert code("/dcgm/device/info/id/abc") == 400   # unparsable id
        assert code("/dcgm/device/info/id/99999999999") == 400
        assert code("/dcgm/device/info/id/7") == 404     # out of range
        assert code("/dcgm/device/info/uuid/GPU-nope") == 404
        assert code("/dcgm/process/info/pid/424242") == 404
        assert code("/dcgm/unknown") == 404
        assert "dcgm_gpu_temp" in get("/metrics") and "gpumon_engine_cpu_percent" in get("/metrics")
        assert "XGMI1" in get("/gpumon/topology")
    finally:
        p.send_signal(signal.SIGTERM)
        p.wait(timeout=10)


def test_start_hostengine_mode_spawns_and_reaps_child():
    E.init(mode=E.START_HOSTENGINE, backend="fake", backend_spec="gpus=2", hostengine_path=bin_path("gpumon-hostengine"))
    out = subprocess.run(["pgrep", "-P", str(os.getpid()), "-f", "gpumon-hostengine"], capture_output=True, text=True)
    pids = [int(x) for x in out.stdout.split()]
    assert len(pids) == 1
    assert os.getpgid(pids[0]) != os.getpgid(0)  # own process group, like the reference's Setpgid
    assert E.device_count() == 2
    assert E.introspect()["memory_kb"] > 0
    E.shutdown()
    time.sleep(0.2)
    assert not os.path.exists(f"/proc/{pids[0]}") or open(f"/proc/{pids[0]}/stat").read().split()[2] == "Z"
    with pytest.raises(E.GpumonError) as e:
        E.init(mode=E.START_HOSTENGINE, hostengine_path="/nonexistent/gpumon-hostengine")
    assert e.value.status in (A.GM_ST_CHILD_SPAWN_FAILED, A.GM_ST_CONNECTION_NOT_VALID)


def test_connect_failure_is_an_error_not_a_hang(tmp_path):
    t = time.time()
    with pytest.raises(E.GpumonError) as e:
        E.init(mode=E.STANDALONE, address=str(tmp_path / "nothing.sock"), is_unix_socket=True)
    assert e.value.status == A.GM_ST_CONNECTION_NOT_VALID and time.time() - t < 15


def test_term_flag_stops_daemon(daemon):
    r = subprocess.run([bin_path("gpumon-hostengine"), "--term", "--pid-file", daemon["pid_file"]], capture_output=True,
                       text=True, timeout=20)
    assert r.returncode == 0 and "terminated" in r.stdout
    daemon["proc"].wait(timeout=5)
    assert not os.path.exists(daemon["sock"])


def test_cli_tools_against_daemon(daemon):
    def run(*args, timeout=30):
        r = subprocess.run([bin_path("gpumon"), *args, "-connect", daemon["sock"], "-socket"], capture_output=True,
                           text=True, timeout=timeout)
        assert r.returncode == 0, r.stderr
        return r.stdout

    assert "UUID           : GPU-00000003-fa4e-0355-0000-000000000000" in run("info")
    assert "P2P Available          : \n    GPU1 - (BusID)0000:15:00.0 - XGMI1" in run("discovery")
    dm = run("dmon", "-c", "2", "-d", "100").splitlines()
    assert dm[0].startswith("# gpu   pwr  temp    sm   mem   enc   dec  mclk  pclk") and len(dm) == 2 + 2 * 4
    assert dm[2].split()[5] == "-"  # no encoder on MI355X
    cols = run("dmon", "-c", "1", "-e", "150,gfx_clock,155,uuid").splitlines()
    assert cols[0].split() == ["#", "gpu", "gpu_temp", "gfx_clock", "power_usage", "uuid"]
    assert "XGMI1" in run("topo") and "Legend" in run("topo")
    assert "Status             : Healthy" in run("health", "-c", "1")
    assert run("introspect", "--wait", "100").startswith("Memory(KB)")
    assert "-" in run("ps", "-c", "1").splitlines()[2]
    out = run("inject", "-g", "1", "-f", "gpu_temp", "-v", "123", "--hold", "2000")
    assert "Successfully injected" in out
    assert run("dmon", "-c", "1", "-e", "150").splitlines()[2].split() == ["1", "123"]
    r = subprocess.run([bin_path("gpumon"), "fields"], capture_output=True, text=True)
    assert "dcgm_sm_clock" in r.stdout and "gpumon_xgmi_link_rx_kb_total" in r.stdout


def test_cli_policy_and_events_embedded():
    env = dict(ENV)
    p = subprocess.Popen([bin_path("gpumon"), "policy", "--conditions", "thermal", "--thermal", "30", "-c", "1",
                          "--timeout", "8000"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    out, err = p.communicate(timeout=20)
    assert p.returncode == 0, err
    assert "Thermal Limit" in out and "ThermalViolation" in out  # the fake idles above 30 C
    r = subprocess.run([bin_path("gpumon"), "pidinfo", "-pid", "1", "--wait", "100"], env=env, capture_output=True,
                       text=True, timeout=20)
    assert r.returncode == 1 and "no GPU activity" in r.stderr


def test_exporter_says_when_the_health_check_cannot_run(tmp_path):
    """An exporter connected to a host engine that goes away: the verdict families disappear (no stale "all healthy")
    and gpumon_health_check_up turns 0, which is what the GpuHealthCheckDown alert reads."""
    import urllib.request
    sock = str(tmp_path / "he.sock")
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2")
    he = subprocess.Popen([bin_path("gpumon-hostengine"), "--domain-socket", sock], env=env, stderr=subprocess.PIPE, text=True)
    ex = None
    try:
        assert "listening" in he.stderr.readline()
        ex = subprocess.Popen([bin_path("gpumon-exporter"), "-connect", sock, "-socket", "--health", "-o", "", "-d", "200", "--listen",
                               "127.0.0.1:0"], stderr=subprocess.PIPE, text=True)
        port = None
        for _ in range(50):
            m = re.search(r"Serving /metrics on port (\d+)", ex.stderr.readline())
            if m:
                port = int(m.group(1))
                break
        assert port

        def body():
            return urllib.request.urlopen(f"http://127.0.0.1:{port}/metrics", timeout=5).read().decode()

        def wait_for(pred):
            deadline = time.time() + 10
            while time.time() < deadline:
                b = body()
                if pred(b):
                    return b
                time.sleep(0.1)
            return body()

        b = wait_for(lambda t: "gpumon_health_check_up 1" in t)
        assert "gpumon_health_check_up 1" in b and b.count("\ngpumon_health_status{") == 16
        he.terminate()
        he.wait(timeout=10)
        b = wait_for(lambda t: "gpumon_health_check_up 0" in t)
        assert "gpumon_health_check_up 0" in b and "gpumon_health_status{" not in b
    finally:
        for p in (ex, he):
            if p and p.poll() is None:
                p.kill()
                p.wait()


def test_term_only_signals_a_host_engine(tmp_path):
    """`gpumon-hostengine --term` reads a pid from a file that may be older than the process it named: a pid that now
    belongs to something else is left alone, and said to be stale."""
    pidfile = tmp_path / "he.pid"
    victim = subprocess.Popen(["sleep", "30"])
    try:
        pidfile.write_text(f"{victim.pid}\n")
        r = subprocess.run([bin_path("gpumon-hostengine"), "--term", "--pid-file", str(pidfile)], capture_output=True, text=True, timeout=30)
        assert r.returncode == 1 and "is not a host engine" in r.stderr and "sleep" in r.stderr, r.stderr
        assert victim.poll() is None                      # still running
    finally:
        victim.kill()
        victim.wait()
    r = subprocess.run([bin_path("gpumon-hostengine"), "--term", "--pid-file", str(pidfile)], capture_output=True, text=True, timeout=30)
    assert r.returncode == 1 and "is gone" in r.stderr, r.stderr


def test_http_listens_on_ipv6_addresses(tmp_path):
    import socket
    import urllib.request
    try:
        s = socket.socket(socket.AF_INET6, socket.SOCK_STREAM)
        s.bind(("::1", 0))
        s.close()
    except OSError:
        pytest.skip("no IPv6 loopback on this host")
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=1")
    ex = subprocess.Popen([bin_path("gpumon-exporter"), "-e", "-o", "", "--listen", "[::1]:0"], stderr=subprocess.PIPE, text=True, env=env)
    try:
        port = None
        for _ in range(50):
            m = re.search(r"Serving /metrics on port (\d+)", ex.stderr.readline())
            if m:
                port = int(m.group(1))
                break
        assert port
        body = urllib.request.urlopen(f"http://[::1]:{port}/metrics", timeout=5).read().decode()
        assert "dcgm_gpu_temp" in body
    finally:
        ex.terminate()
        ex.wait(timeout=10)
