"""Deployment artefacts stay consistent with the binaries: manifests parse, the
flags they pass are flags the tools accept, ports / paths / scrape intervals are
the contract values (exporters/prometheus-dcgm/k8s/*.yaml, docker/*, bare-metal/*
of the reference, re-targeted at AMD device nodes)."""
import glob
import json
import os
import re
import subprocess

import yaml

from conftest import bin_path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEPLOY = os.path.join(ROOT, "deploy")


def _docs(path):
    with open(path) as f:
        return [d for d in yaml.safe_load_all(f) if d]


def test_all_yaml_parses():
    files = glob.glob(os.path.join(DEPLOY, "**", "*.y*ml"), recursive=True) + [os.path.join(ROOT, ".github/workflows/ci.yml")]
    assert len(files) >= 8
    for f in files:
        assert _docs(f), f


def test_daemonsets_target_amd_nodes_and_share_memory_volumes():
    sets = [d for f in glob.glob(os.path.join(DEPLOY, "k8s", "**", "*daemonset.yaml"), recursive=True) for d in _docs(f)]
    assert len(sets) == 4
    for ds in sets:
        spec = ds["spec"]["template"]["spec"]
        assert ds["kind"] == "DaemonSet" and spec["nodeSelector"] == {"hardware-type": "AMDGPU"}
        names = {c["name"] for c in spec["containers"]}
        assert "gpumon-exporter" in names
        exporter = next(c for c in spec["containers"] if c["name"] == "gpumon-exporter")
        mounts = {m["mountPath"] for m in exporter["volumeMounts"]}
        assert {"/dev/kfd", "/dev/dri", "/run/prometheus"} <= mounts
        shared = [v for v in spec["volumes"] if v.get("emptyDir", {}).get("medium") == "Memory"]
        assert shared, "metrics travel through an in-memory volume"
        if "pod-gpu-metrics-exporter" in names:
            pod = next(c for c in spec["containers"] if c["name"] == "pod-gpu-metrics-exporter")
            assert any(p["containerPort"] == 9400 for p in pod["ports"])
            assert "/var/lib/kubelet/pod-resources" in {m["mountPath"] for m in pod["volumeMounts"]}


def test_prometheus_scrapes_gpu_metrics_every_second():
    cm = _docs(os.path.join(DEPLOY, "k8s/pod-gpu-metrics-exporter/prometheus/prometheus-configmap.yaml"))[0]
    job = yaml.safe_load(cm["data"]["prometheus.yml"])["scrape_configs"][0]
    assert job["job_name"] == "gpu_metrics" and job["scrape_interval"] == "1s" and job["metrics_path"] == "/gpu/metrics"
    compose_prom = _docs(os.path.join(DEPLOY, "docker/prometheus/prometheus.yml"))[0]
    fast = {j["job_name"]: j.get("scrape_interval") for j in compose_prom["scrape_configs"]}
    assert fast["node_exporter"] == "1s" and fast["gpumon_exporter"] == "1s"
    compose = _docs(os.path.join(DEPLOY, "docker/docker-compose.yml"))[0]
    assert compose["services"]["gpumon_exporter"]["devices"] == ["/dev/kfd", "/dev/dri"]
    assert compose["volumes"]["collector_data"]["driver_opts"]["type"] == "tmpfs"


def _entrypoint(dockerfile):
    text = open(os.path.join(DEPLOY, "docker", dockerfile)).read()
    return eval(re.search(r"^ENTRYPOINT (\[.*\])$", text, re.M).group(1))  # a JSON list literal


def test_image_entrypoints_use_flags_the_binaries_accept(tmp_path):
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2")
    ep = _entrypoint("gpumon-exporter.Dockerfile")
    assert ep[0] == "gpumon-exporter" and "-e" in ep
    out = tmp_path / "dcgm.prom"
    args = [a if a != ":9400" else "127.0.0.1:0" for a in ep[1:]]
    r = subprocess.run([bin_path(ep[0]), *args, "-o", str(out), "--once"], env=env, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    assert "dcgm_gpu_temp" in out.read_text()
    ep = _entrypoint("pod-exporter.Dockerfile")
    r = subprocess.run([bin_path(ep[0]), *ep[1:], "--source", str(tmp_path / "x.prom"), "--dest", str(tmp_path / "y.prom"),
                        "--listen", "", "--liveness-ms", "300"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "usage" not in r.stderr  # flags accepted; exits through the liveness watchdog


def test_systemd_units_reference_installed_binaries():
    for unit in ("gpumon-exporter.service", "gpumon-hostengine.service"):
        text = open(os.path.join(DEPLOY, "bare-metal", unit)).read()
        exe = re.search(r"^ExecStart=(\S+)", text, re.M).group(1)
        assert os.path.exists(bin_path(os.path.basename(exe)))
        assert "Restart=always" in text
    exporter = open(os.path.join(DEPLOY, "bare-metal/gpumon-exporter.service")).read()
    assert "After=gpumon-hostengine.service" in exporter and "/run/prometheus/dcgm.prom" in exporter
    # the flags of the unit files parse: --help style dry run is not available, so check against the usage text
    mk = open(os.path.join(DEPLOY, "bare-metal/Makefile")).read()
    assert "gpumon-hostengine" in mk and "systemctl enable --now" in mk


def test_alert_rules_only_name_families_the_exporter_produces(tmp_path):
    """Every metric an alert expression reads exists: in the exporter's output on the scripted backend with the flags
    the deployments pass, or among the diagnostics families of core/exporter/diag_results.cc."""
    rules_path = os.path.join(DEPLOY, "docker", "prometheus", "gpumon-alerts.yml")
    rules = yaml.safe_load(open(rules_path))
    alerts = [r for g in rules["groups"] for r in g["rules"]]
    assert len(alerts) >= 15 and len({r["alert"] for r in alerts}) == len(alerts)
    assert all(r["labels"]["severity"] in ("warning", "critical") and "summary" in r["annotations"] for r in alerts)
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2")
    r = subprocess.run([bin_path("gpumon-exporter"), "-e", "-p", "--health", "--extras", "--self-metrics", "--once", "-o", ""], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    produced = set(re.findall(r"^# TYPE (\w+) ", r.stdout, re.M))
    # the diagnostics families come from a report file, not from the backend: their names are read from the source
    produced |= set(re.findall(r'"(gpumon_diag_\w+)"', open(os.path.join(ROOT, "core", "exporter", "diag_results.cc")).read()))
    functions = {"increase", "rate", "changes", "time", "timestamp", "and", "or", "for", "level", "failure"}
    for a in alerts:
        names = set(re.findall(r"\b([a-z_][a-z0-9_]*)\b", re.sub(r'"[^"]*"', "", a["expr"]))) - functions
        names = {n for n in names if not re.fullmatch(r"\d+[smh]", n)}
        assert names and names <= produced, (a["alert"], sorted(names - produced))
    # the Prometheus configurations load the file, and the Kubernetes ConfigMap carries the same rules
    prom = yaml.safe_load(open(os.path.join(DEPLOY, "docker", "prometheus", "prometheus.yml")))
    assert prom["rule_files"] == ["/etc/prometheus/gpumon-alerts.yml"]
    assert "gpumon-alerts.yml" in open(os.path.join(DEPLOY, "docker", "prometheus", "Dockerfile")).read()
    cm = yaml.safe_load(open(os.path.join(DEPLOY, "k8s", "pod-gpu-metrics-exporter", "prometheus", "prometheus-alerts-configmap.yaml")))
    assert yaml.safe_load(cm["data"]["gpumon-alerts.yml"]) == rules
    k8s_prom = yaml.safe_load(open(os.path.join(DEPLOY, "k8s", "pod-gpu-metrics-exporter", "prometheus", "prometheus-configmap.yaml")))
    assert yaml.safe_load(k8s_prom["data"]["prometheus.yml"])["rule_files"] == ["/etc/prometheus/rules/gpumon-alerts.yml"]


def test_grafana_dashboard_queries_families_the_exporter_produces():
    dash = json.load(open(os.path.join(DEPLOY, "docker", "grafana", "dashboards", "gpumon-node.json")))
    assert dash["uid"] and len(dash["panels"]) >= 12 and len({p["id"] for p in dash["panels"]}) == len(dash["panels"])
    env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2")
    r = subprocess.run([bin_path("gpumon-exporter"), "-e", "-p", "--health", "--extras", "--self-metrics", "--once", "-o", ""], env=env,
                       capture_output=True, text=True, timeout=60)
    assert r.returncode == 0, r.stderr
    produced = set(re.findall(r"^# TYPE (\w+) ", r.stdout, re.M))
    produced |= set(re.findall(r'"(gpumon_diag_\w+)"', open(os.path.join(ROOT, "core", "exporter", "diag_results.cc")).read()))
    queried = set()
    for p in dash["panels"]:
        assert p["datasource"]["uid"] == "prometheus" and p["targets"]
        for t in p["targets"]:
            queried |= set(re.findall(r"\b((?:dcgm|gpumon)_\w+)", t["expr"]))
    assert len(queried) >= 20 and queried <= produced, sorted(queried - produced)
    # the compose stack provisions the data source the panels name, and mounts the dashboard directory
    ds = yaml.safe_load(open(os.path.join(DEPLOY, "docker", "grafana", "provisioning", "datasources", "prometheus.yml")))
    assert ds["datasources"][0]["uid"] == "prometheus"
    compose = yaml.safe_load(open(os.path.join(DEPLOY, "docker", "docker-compose.yml")))
    mounts = compose["services"]["grafana"]["volumes"]
    assert any(m.startswith("./grafana/dashboards:") for m in mounts) and any(m.startswith("./grafana/provisioning:") for m in mounts)
    assert compose["services"]["gpumon_exporter"]["command"] == ["--health", "--extras", "--self-metrics"]


def test_one_listener_per_port_in_every_pod(tmp_path):
    """Containers of a pod share a network namespace: where the pod exporter serves :9400, the node exporter image
    (whose entrypoint also listens on :9400) must be told not to, and the flags it is given must be accepted."""
    ep = _entrypoint("gpumon-exporter.Dockerfile")
    assert ep[-2:] == ["--listen", ":9400"]
    checked = 0
    for path in sorted(glob.glob(os.path.join(DEPLOY, "k8s", "**", "*.yaml"), recursive=True)):
        for doc in yaml.safe_load_all(open(path)):
            if not doc or doc.get("kind") != "DaemonSet":
                continue
            containers = doc["spec"]["template"]["spec"]["containers"]
            images = {c["image"].split(":")[0]: c for c in containers}
            if "gpumon/exporter" not in images:
                continue
            args = ep[1:] + images["gpumon/exporter"].get("args", [])
            listen = [args[i + 1] for i, a in enumerate(args) if a == "--listen"][-1]  # the last one wins
            if "gpumon/pod-exporter" in images:
                assert listen == "", path
            # the whole command line parses and produces a file
            out = tmp_path / f"{checked}.prom"
            argv = [a for a in args]
            if "-o" in argv:
                argv[argv.index("-o") + 1] = str(out)
            else:
                argv += ["-o", str(out)]
            argv = [a if a != ":9400" else "127.0.0.1:0" for a in argv]
            env = dict(os.environ, GPUMON_BACKEND="fake", GPUMON_BACKEND_SPEC="gpus=2")
            r = subprocess.run([bin_path("gpumon-exporter"), *argv, "--once"], env=env, capture_output=True, text=True, timeout=60)
            assert r.returncode == 0 and "gpumon_health_overall" in out.read_text(), (path, r.stderr)
            checked += 1
    assert checked == 4


def test_all_gpu_diagnostics_wrapper(tmp_path):
    """deploy/docker/gpumon-diag-all.sh with a stand-in for gpumon-diag: one run that loads all GPUs at once (or one run
    per GPU when asked for), the fabric tests once and only with more than one GPU, reports in the directory, the worst
    exit status wins, nothing is started after a step that died, flags are ones gpumon-diag knows."""
    script = os.path.join(DEPLOY, "docker", "gpumon-diag-all.sh")
    stub = tmp_path / "gpumon-diag"
    stub.write_text("""#!/bin/bash
echo "$@" >> "$STUB_LOG"
if [ "$1" = info ]; then
  for ((i = 0; i < STUB_GPUS; i++)); do echo "GPU $i: AMD Instinct MI355X (gfx950) 0000:0$i:00.0  CUs 256"; done
  [ "$STUB_GPUS" -gt 0 ]; exit $?
fi
prev=""
for a in "$@"; do [ "$prev" = --publish ] && echo '{}' > "$a"; prev=$a; done
[ "$1" = run ] && [ "$4" = "-g" ] && [ "$5" = "${STUB_FAIL_GPU:-none}" ] && exit 1
[ "$1" = run ] && [ "$4" = "-g" ] && [ "$5" = "${STUB_CRASH_GPU:-none}" ] && kill -SEGV $$
[ "$1" = run ] && [ "$4" = "--gpus" ] && [ -n "${STUB_FAIL_NODE:-}" ] && exit 1
[ "$1" = run ] && [ "$4" = "--gpus" ] && [ -n "${STUB_CRASH_NODE:-}" ] && kill -SEGV $$
exit 0
""")
    stub.chmod(0o755)

    def run(gpus, **env):
        log, out = tmp_path / f"log{gpus}{len(env)}", tmp_path / f"out{gpus}{len(env)}"
        e = dict(os.environ, GPUMON_DIAG_BIN=str(stub), GPUMON_DIAG_DIR=str(out), STUB_LOG=str(log), STUB_GPUS=str(gpus), **env)
        r = subprocess.run(["bash", script], env=e, capture_output=True, text=True, timeout=30)
        calls = log.read_text().splitlines() if log.exists() else []
        return r.returncode, calls, sorted(p.name for p in out.iterdir()) if out.exists() else []

    rc, calls, files = run(1)
    assert rc == 0 and files == ["node.json", "preflight.json"]
    assert calls == ["info", f"preflight --publish {tmp_path}/out10/preflight.json", f"run -r 2 --gpus all --no-fabric --publish {tmp_path}/out10/node.json"]
    # all GPUs at once; the fabric tests once
    rc, calls, files = run(8, GPUMON_DIAG_LEVEL="3", GPUMON_DIAG_SUBMIT="127.0.0.1:5555", STUB_FAIL_NODE="1")
    assert rc == 1 and len(calls) == 5 and files == ["links.json", "node.json", "preflight.json", "rccl.json"]
    assert calls[1].startswith("preflight --publish") and calls[1].endswith("--submit 127.0.0.1:5555")  # the deployment checks come first
    assert calls[2].startswith("run -r 3 --gpus all --no-fabric --publish") and calls[2].endswith("--submit 127.0.0.1:5555")
    assert calls[3].startswith("links --publish") and calls[4].startswith("rccl --scaling --publish")
    rc, calls, files = run(2, X="1")
    assert rc == 0 and calls[4].startswith("rccl --max-bytes 256M --factor 16 --publish")
    rc, calls, files = run(8, GPUMON_DIAG_EXPECT_GPUS="8", X="1", Y="2", Z="3")
    assert calls[1].startswith("preflight --expect-gpus 8 --publish")
    # one GPU after the other, when asked for
    rc, calls, files = run(3, GPUMON_DIAG_LEVEL="1", GPUMON_DIAG_SEQUENTIAL="1", STUB_FAIL_GPU="1")
    assert rc == 1 and len(calls) == 2 + 3 + 2 and files == ["gpu0.json", "gpu1.json", "gpu2.json", "links.json", "preflight.json", "rccl.json"]
    assert calls[3].startswith("run -r 1 -g 1 --no-fabric ")
    assert run(0)[0] == 2
    rc, calls, files = run(4, GPUMON_DIAG_SEQUENTIAL="1", STUB_CRASH_GPU="1", X="1", Y="2")  # GPU 1's run dies: GPUs 2 and 3, links and RCCL are not started
    assert rc == 2 and [c.split()[0:5] for c in calls[2:]] == [["run", "-r", "2", "-g", "0"], ["run", "-r", "2", "-g", "1"]]
    rc, calls, files = run(4, STUB_CRASH_NODE="1", X="1")      # the concurrent run dies: no fabric test is started
    assert rc == 2 and len(calls) == 3
    # every flag the script passes is one the real binary parses
    src = open(os.path.join(ROOT, "cmd", "gpumon-diag.cc")).read()
    for flag in ("-r", "-g", "--gpus", "--publish", "--submit", "--max-bytes", "--factor", "--expect-gpus"):
        assert f'"{flag}"' in src, flag
    for flag in ("--no-fabric", "--scaling"):
        assert f'"{flag}"' in src, flag
    ep = _entrypoint("gpumon-diag.Dockerfile")
    assert ep == ["gpumon-diag-all"] and "gpumon-diag-all.sh /usr/local/bin/gpumon-diag-all" in open(
        os.path.join(DEPLOY, "docker", "gpumon-diag.Dockerfile")).read()
    cron = yaml.safe_load(open(os.path.join(DEPLOY, "k8s", "diag", "gpumon-diag-cronjob.yaml")))
    pod = cron["spec"]["jobTemplate"]["spec"]["template"]["spec"]
    assert cron["kind"] == "CronJob" and pod["restartPolicy"] == "Never" and pod["nodeSelector"] == {"hardware-type": "AMDGPU"}
    assert pod["containers"][0]["resources"]["limits"] == {"amd.com/gpu": 8}
    env = {e["name"]: e["value"] for e in pod["containers"][0]["env"]}
    assert env["GPUMON_DIAG_EXPECT_GPUS"] == "8"        # a GPU that fell off the bus fails the deployment checks
    assert pod["volumes"][0]["hostPath"]["path"] == "/var/lib/gpumon/diag"


def test_runbook_covers_every_alert_and_names_real_commands():
    """docs/runbook.md has a row for every alert of the rule file, and the sub-commands it tells an operator to run exist."""
    import re
    import subprocess
    rules = open(os.path.join(ROOT, "deploy", "docker", "prometheus", "gpumon-alerts.yml")).read()
    book = open(os.path.join(ROOT, "docs", "runbook.md")).read()
    alerts = re.findall(r"- alert: (\w+)", rules)
    assert len(alerts) >= 20
    missing = [a for a in alerts if f"`{a}`" not in book]
    assert not missing, missing
    usage = subprocess.run([os.path.join(ROOT, "build", "bin", "gpumon")], capture_output=True, text=True).stderr
    known = set(re.search(r"<([a-z|]+)>", usage).group(1).split("|"))
    for cmd in set(re.findall(r"`gpumon (\w+)", book)):
        assert cmd in known, cmd
    diag_usage = subprocess.run([os.path.join(ROOT, "build", "bin", "gpumon-diag")], capture_output=True, text=True).stderr
    diag_known = set(re.search(r"<([a-z|]+)>", diag_usage).group(1).split("|"))
    for cmd in set(re.findall(r"`gpumon-diag (\w+)", book)):
        assert cmd in diag_known, cmd
