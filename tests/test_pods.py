"""Pod attribution: the relabeler, the kubelet PodResources client (checked
against a real grpcio server speaking HTTP/2), the device-plugin checkpoint
fallback and the file-watch pod exporter.

Reference behaviour: exporters/prometheus-dcgm/k8s/pod-gpu-metrics-exporter/src
(device_pod.go:26-113 rewriter, kubelet_server.go:20-53 client, watch.go:11-48,
http.go:13-44).  The reference has no tests for any of it."""
import json
import os
import socket
import subprocess
import time
import urllib.error
import urllib.request
from concurrent import futures

import grpc
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from conftest import bin_path
from k8s_gpu_monitor_amd import engine as E
from k8s_gpu_monitor_amd.utils import promparse

SAMPLE = """# HELP dcgm_gpu_temp GPU temperature (in C).
# TYPE dcgm_gpu_temp gauge
dcgm_gpu_temp{gpu="0",uuid="GPU-aaaa"} 41
dcgm_gpu_temp{gpu="1",uuid="GPU-bbbb"} 52
dcgm_gpu_temp{gpu="2",uuid="GPU-cccc"} 63
"""


# ------------------------------------------------------------- protobuf by hand --
def _varint(n):
    out = b""
    while True:
        b = n & 0x7F
        n >>= 7
        out += bytes([b | (0x80 if n else 0)])
        if not n:
            return out


def _ld(field, payload):
    if isinstance(payload, str):
        payload = payload.encode()
    return _varint(field << 3 | 2) + _varint(len(payload)) + payload


def pod_resources_response(pods):
    """pods: [(name, namespace, [(container, [(resource, [ids])])])] -> ListPodResourcesResponse bytes."""
    out = b""
    for name, ns, containers in pods:
        pod = _ld(1, name) + _ld(2, ns)
        for cname, devices in containers:
            c = _ld(1, cname)
            for resource, ids in devices:
                d = _ld(1, resource) + b"".join(_ld(2, i) for i in ids)
                c += _ld(2, d)
            pod += _ld(3, c)
        out += _ld(1, pod)
    return out


class FakeKubelet:
    """A PodResourcesLister served by grpcio on a unix socket."""

    def __init__(self, path, pods, versions=("v1", "v1alpha1"), delay=0.0):
        self.path, self.pods, self.calls, self.delay = path, pods, [], delay
        self.server = grpc.server(futures.ThreadPoolExecutor(max_workers=2))
        for v in versions:
            def handler(request, context, v=v):
                self.calls.append(v)
                if self.delay:
                    time.sleep(self.delay)
                return pod_resources_response(self.pods)
            self.server.add_generic_rpc_handlers((grpc.method_handlers_generic_handler(
                f"{v}.PodResourcesLister", {"List": grpc.unary_unary_rpc_method_handler(handler)}),))
        self.server.add_insecure_port(f"unix:{path}")
        self.server.start()

    def stop(self):
        self.server.stop(0)


PODS = [
    ("trainer-0", "ml", [("main", [("amd.com/gpu", ["GPU-aaaa", "GPU-bbbb"])]),
                         ("sidecar", [("example.com/nic", ["nic0"])])]),
    ("no-gpu-pod", "default", [("web", [])]),
    ("infer-7", "serving", [("srv", [("amd.com/gpu", ["amdgpu2"])])]),
]


@pytest.fixture
def kubelet(tmp_path):
    k = FakeKubelet(str(tmp_path / "kubelet.sock"), PODS)
    yield k
    k.stop()


# ----------------------------------------------------------------- relabeler --
def test_relabel_by_uuid_and_by_index():
    out = E.pod_relabel(SAMPLE, {"GPU-aaaa": ("p1", "ns1", "c1"), "amdgpu2": ("p2", "ns2", "c2")})
    lines = out.splitlines()
    assert lines[:2] == SAMPLE.splitlines()[:2]  # HELP / TYPE untouched
    assert lines[2] == 'dcgm_gpu_temp{gpu="0",uuid="GPU-aaaa",pod_name="p1",pod_namespace="ns1",container_name="c1"} 41'
    assert lines[3] == 'dcgm_gpu_temp{gpu="1",uuid="GPU-bbbb"} 52'  # unowned GPUs pass through
    assert lines[4] == 'dcgm_gpu_temp{gpu="2",uuid="GPU-cccc",pod_name="p2",pod_namespace="ns2",container_name="c2"} 63'
    assert out.endswith("\n")
    assert E.pod_relabel(out, {"GPU-aaaa": ("p1", "ns1", "c1"), "amdgpu2": ("p2", "ns2", "c2")}) == out  # idempotent
    # the reference's device-plugin naming also works by changing the prefix
    assert 'pod_name="x"' in E.pod_relabel(SAMPLE, {"nvidia1": ("x", "y", "z")}, index_prefix="nvidia").splitlines()[3]


def test_relabel_survives_hostile_label_values():
    """The reference splits on ',' and '}' (device_pod.go:80-97) and corrupts such lines; we parse."""
    text = 'm{gpu="0",uuid="GPU-a,b}c",model="MI355X \\"OAM\\", rev}2"} 1\nbroken{gpu="0" 1\nno_labels 7\n\n'
    out = E.pod_relabel(text, {"GPU-a,b}c": ('po"d', "n\\s", "c x")})
    l0 = out.splitlines()[0]
    assert l0.startswith('m{gpu="0",uuid="GPU-a,b}c",model="MI355X \\"OAM\\", rev}2",pod_name="po\\"d",')
    assert 'pod_namespace="n\\\\s",container_name="c x"} 1' in l0
    assert out.splitlines()[1:] == ['broken{gpu="0" 1', "no_labels 7", ""]
    fam = promparse.parse(l0 + "\n", require_type=False)
    labels = fam["m"].samples[0][0]
    assert labels["pod_name"] == 'po"d' and labels["pod_namespace"] == "n\\s" and labels["uuid"] == "GPU-a,b}c"


_label_value = st.text(alphabet=st.characters(blacklist_categories=("Cs",), blacklist_characters="\x00"), max_size=12)


@settings(max_examples=150, deadline=None)
@given(uuids=st.lists(st.text(alphabet="abcdef0123456789-,}{\"\\ =", min_size=1, max_size=10), min_size=1, max_size=6,
                      unique=True),
       owned=st.lists(st.booleans(), min_size=6, max_size=6), pod=_label_value, ns=_label_value, cont=_label_value,
       extra=_label_value)
def test_relabel_properties(uuids, owned, pod, ns, cont, extra):
    """For any label contents: output still parses; values and original labels are unchanged;
    exactly the owned GPUs gain the three labels; relabelling twice changes nothing."""
    pod, ns, cont = (x.replace("\t", " ").replace("\n", " ").replace("\r", " ") for x in (pod, ns, cont))  # TSV map input
    esc = lambda v: v.replace("\\", "\\\\").replace('"', '\\"').replace("\n", "\\n")
    text = "# TYPE t gauge\n" + "".join(
        f't{{gpu="{i}",uuid="{esc(u)}",x="{esc(extra)}"}} {i * 1.5}\n' for i, u in enumerate(uuids))
    pmap = {u: (pod, ns, cont) for i, u in enumerate(uuids) if owned[i] and u.strip() == u and "\t" not in u}
    out = E.pod_relabel(text, pmap)
    before, after = promparse.parse(text)["t"].samples, promparse.parse(out)["t"].samples
    assert len(before) == len(after)
    for (bl, bv), (al, av) in zip(before, after):
        assert av == bv
        assert {k: v for k, v in al.items() if k in bl} == bl
        if bl["uuid"] in pmap:
            assert (al["pod_name"], al["pod_namespace"], al["container_name"]) == (pod, ns, cont)
        else:
            assert "pod_name" not in al
    assert E.pod_relabel(out, pmap) == out


# ------------------------------------------------------------ kubelet client --
def test_kubelet_client_v1(kubelet):
    m = E.kubelet_pod_map(kubelet.path)
    assert m == {"GPU-aaaa": ("trainer-0", "ml", "main"), "GPU-bbbb": ("trainer-0", "ml", "main"),
                 "amdgpu2": ("infer-7", "serving", "srv")}
    assert kubelet.calls == ["v1"]
    # several resource names: the reference hard-codes one (device_pod.go:14)
    m = E.kubelet_pod_map(kubelet.path, "amd.com/gpu,example.com/nic")
    assert m["nic0"] == ("trainer-0", "ml", "sidecar")


def test_kubelet_client_falls_back_to_v1alpha1(tmp_path):
    k = FakeKubelet(str(tmp_path / "old.sock"), PODS, versions=("v1alpha1",))
    try:
        assert len(E.kubelet_pod_map(k.path)) == 3
        assert k.calls == ["v1alpha1"]
    finally:
        k.stop()


def test_kubelet_client_large_response(tmp_path):
    """Bigger than one HTTP/2 frame and than the default flow-control window."""
    pods = [(f"pod-{i:05d}", "bulk", [("c", [("amd.com/gpu", [f"GPU-{i:05d}-{j}" for j in range(8)])])]) for i in range(3000)]
    k = FakeKubelet(str(tmp_path / "big.sock"), pods)
    try:
        m = E.kubelet_pod_map(k.path)
        assert len(m) == 24000 and m["GPU-02999-7"] == ("pod-02999", "bulk", "c")
    finally:
        k.stop()


def test_kubelet_client_errors(tmp_path):
    with pytest.raises(E.GpumonError):
        E.kubelet_pod_map(str(tmp_path / "absent.sock"), timeout_ms=500)
    # a listener that never answers: the deadline (kubelet_server.go:14-16 uses 10 s) must fire
    s = socket.socket(socket.AF_UNIX)
    s.bind(str(tmp_path / "mute.sock"))
    s.listen(1)
    t = time.time()
    with pytest.raises(E.GpumonError):
        E.kubelet_pod_map(str(tmp_path / "mute.sock"), timeout_ms=400)
    assert time.time() - t < 5
    s.close()
    k = FakeKubelet(str(tmp_path / "none.sock"), PODS, versions=())  # neither API version served
    try:
        with pytest.raises(E.GpumonError):
            E.kubelet_pod_map(k.path, timeout_ms=2000)
    finally:
        k.stop()


# ----------------------------------------------------- exporter integrations --
def test_exporter_attaches_pod_labels_from_kubelet(tmp_path):
    E.init(backend="fake", backend_spec="gpus=3,seed=1")
    try:
        uuid0 = E.device_attributes(0)["uuid"]
        k = FakeKubelet(str(tmp_path / "k.sock"), [("job", "team", [("c", [("amd.com/gpu", [uuid0, "amdgpu2"])])])])
        try:
            with E.Exporter(kubelet_socket=k.path) as ex:
                ex.sweep()
                fam = promparse.parse(ex.render())
            by_gpu = {l["gpu"]: l for l, _ in fam["dcgm_gpu_temp"].samples}
            assert by_gpu["0"]["pod_name"] == "job" and by_gpu["0"]["pod_namespace"] == "team"
            assert "pod_name" not in by_gpu["1"]
            assert by_gpu["2"]["container_name"] == "c"
        finally:
            k.stop()
        # kubelet gone: the sweep still publishes, without pod labels
        with E.Exporter(kubelet_socket=str(tmp_path / "gone.sock")) as ex:
            ex.sweep()
            assert "pod_name" not in ex.render() and "dcgm_gpu_temp" in ex.render()
    finally:
        E.shutdown()


def test_a_kubelet_that_does_not_answer_never_holds_up_a_sweep(tmp_path):
    """The pod map has a thread and a kept connection of its own (the reference dials the kubelet once a second on the
    path that publishes: src/watchers.go:41 -> kubelet_server.go:36-41).  With a kubelet that takes 3 s per answer the
    file is still published every interval, the labels arrive when the answer does, and all queries share one
    connection; when the kubelet restarts, the next query opens a second one."""
    E.init(backend="fake", backend_spec="gpus=2,seed=1")
    try:
        uuid0 = E.device_attributes(0)["uuid"]
        pods = [("job", "team", [("c", [("amd.com/gpu", [uuid0])])])]
        k = FakeKubelet(str(tmp_path / "k.sock"), pods)
        out = tmp_path / "m.prom"
        ex = E.Exporter(kubelet_socket=k.path, output_path=str(out), interval_ms=100, pod_refresh_ms=300, self_metrics=True)
        try:
            assert len(k.calls) == 1                      # the start-up query, answered at once
            k.delay = 3.0
            ex.start()
            stamps = set()
            t_end = time.time() + 2.0
            while time.time() < t_end:
                if out.exists():
                    stamps.add(out.stat().st_mtime_ns)
                time.sleep(0.02)
            assert len(stamps) >= 12, len(stamps)         # 20 sweeps due in 2 s; a blocked loop would show 1 or 2
            assert len(k.calls) == 2                      # one query in flight the whole time
            k.delay = 0.0
            time.sleep(2.2)                               # the slow answer arrives, quick ones follow
            text = out.read_text()
            assert 'pod_name="job"' in text
            fam = promparse.parse(text)
            assert fam["gpumon_exporter_kubelet_connects_total"].samples[0][1] == 1
            assert fam["gpumon_exporter_pod_map_refreshes_total"].samples[0][1] >= 4
            assert fam["gpumon_exporter_pod_map_up"].samples[0][1] == 1
            # kubelet restart: the kept connection is dead, the next query replaces it without losing the map
            k.stop()
            time.sleep(0.5)
            k = FakeKubelet(k.path, pods)
            time.sleep(1.0)
            fam = promparse.parse(out.read_text())
            assert fam["gpumon_exporter_kubelet_connects_total"].samples[0][1] >= 2
            assert fam["gpumon_exporter_pod_map_up"].samples[0][1] == 1 and 'pod_name="job"' in out.read_text()
        finally:
            ex.stop()
            ex.close()
            k.stop()
    finally:
        E.shutdown()


def test_exporter_checkpoint_file_fallback(tmp_path):
    E.init(backend="fake", backend_spec="gpus=2,seed=1")
    try:
        uuid1 = E.device_attributes(1)["uuid"]
        ck = {"Data": {"PodDeviceEntries": [
            {"PodUID": "uid-1", "ContainerName": "c1", "ResourceName": "amd.com/gpu", "DeviceIDs": {"0": [uuid1]}},
            {"PodUID": "uid-2", "ContainerName": "c2", "ResourceName": "amd.com/gpu", "DeviceIDs": ["amdgpu0"]},
            {"PodUID": "uid-3", "ContainerName": "c3", "ResourceName": "other/thing", "DeviceIDs": ["amdgpu1"]}],
            "RegisteredDevices": {}}, "Checksum": 1}
        path = tmp_path / "kubelet_internal_checkpoint"
        path.write_text(json.dumps(ck))
        with E.Exporter(pod_map_path=str(path)) as ex:
            ex.sweep()
            fam = promparse.parse(ex.render())
        by_gpu = {l["gpu"]: l for l, _ in fam["dcgm_power_usage"].samples}
        assert by_gpu["1"]["pod_name"] == "uid-1" and by_gpu["1"]["container_name"] == "c1"
        assert by_gpu["0"]["pod_name"] == "uid-2" and by_gpu["0"]["pod_namespace"] == ""
    finally:
        E.shutdown()


def _get(port, path="/gpu/metrics"):
    return urllib.request.urlopen(f"http://127.0.0.1:{port}{path}", timeout=5).read().decode()


def test_pod_exporter_file_watch_end_to_end(tmp_path, kubelet):
    src, dst = tmp_path / "run" / "dcgm.prom", tmp_path / "out" / "dcgm-pod.prom"
    src.parent.mkdir()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    p = subprocess.Popen([bin_path("gpumon-pod-exporter"), "--source", str(src), "--dest", str(dst), "--kubelet-socket",
                          kubelet.path, "--listen", f"127.0.0.1:{port}", "--liveness-ms", "60000"],
                         stderr=subprocess.PIPE, text=True)
    try:
        deadline = time.time() + 10
        while time.time() < deadline:
            try:
                _get(port)
            except urllib.error.HTTPError as e:
                assert e.code == 500  # nothing published yet: the reference answers 500 too (http.go:35-39)
                break
            except OSError:
                time.sleep(0.05)
        # the collector publishes by write-to-temp + rename, as the reference's does
        for round_, temp in enumerate((41, 77)):
            tmp = src.with_suffix(".swp")
            tmp.write_text(SAMPLE.replace(" 41", f" {temp}"))
            os.rename(tmp, src)
            want = f'dcgm_gpu_temp{{gpu="0",uuid="GPU-aaaa",pod_name="trainer-0",pod_namespace="ml",container_name="main"}} {temp}'
            deadline = time.time() + 10
            body = ""
            while time.time() < deadline:
                try:
                    body = _get(port)
                    if want in body:
                        break
                except urllib.error.HTTPError:
                    pass
                time.sleep(0.05)
            assert want in body, body
            assert 'uuid="GPU-cccc",pod_name="infer-7"' in body
            assert dst.read_text() == body
        assert len(kubelet.calls) >= 2  # one kubelet query per published file
        with pytest.raises(urllib.error.HTTPError) as e:
            _get(port, "/other")
        assert e.value.code == 404
    finally:
        p.terminate()
        p.wait(timeout=10)
    assert p.returncode == 0 and "Exiting pod exporter" in p.stderr.read()


def test_pod_exporter_liveness_watchdog(tmp_path):
    """No file event within the window -> exit non-zero, like the reference's 10 min wait (watch.go:44-46)."""
    (tmp_path / "w").mkdir()
    p = subprocess.run([bin_path("gpumon-pod-exporter"), "--source", str(tmp_path / "w" / "dcgm.prom"), "--dest",
                        str(tmp_path / "w" / "out.prom"), "--listen", "", "--liveness-ms", "600", "--kubelet-socket",
                        str(tmp_path / "none.sock")], capture_output=True, text=True, timeout=20)
    assert p.returncode == 1 and "too slow" in p.stderr
