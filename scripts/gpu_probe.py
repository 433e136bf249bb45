"""First-contact probe for a real MI355X: device layer vs amd-smi, sampler cost,
and every diagnostic kernel once.  Writes gpurun_out/probe.json."""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = {}
os.makedirs("gpurun_out", exist_ok=True)

def sh(cmd, timeout=60):
    try:
        r = subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=timeout)
        return {"rc": r.returncode, "out": r.stdout[-6000:], "err": r.stderr[-2000:]}
    except Exception as e:
        return {"rc": -1, "err": repr(e)}

which = sys.argv[1:] or ["smi", "engine", "diag"]
if "smi" in which:
    out["amd_smi_static"] = sh("amd-smi static --json 2>&1 | head -c 20000")
    out["amd_smi_metric"] = sh("amd-smi metric --json 2>&1 | head -c 20000")
    out["amd_smi_list"] = sh("amd-smi list --json")
    out["rocm_smi"] = sh("rocm-smi --showuse --showmemuse --showpower --showtemp --showclocks --json")

if "engine" in which:
    from k8s_gpu_monitor_amd import engine as E, abi as A
    t0 = time.time()
    E.init(backend="amdsmi")
    out["init_s"] = time.time() - t0
    out["count"] = E.device_count()
    out["attrs"] = [E.device_attributes(i) for i in range(out["count"])]
    out["status"] = [E.device_status(i) for i in range(out["count"])]
    out["procs"] = [E.device_processes(i) for i in range(out["count"])]
    out["topology"] = [E.device_topology(i) for i in range(out["count"])]
    out["health"] = [E.health_check_gpu(i) for i in range(out["count"])]
    fields = E.field_list()
    g = E.group_create("probe", list(range(out["count"])))
    fg = E.field_group_create("probe", [f for f in fields if not E.field_meta(f)["is_static"]])
    E.watch_fields(g, fg, 100000, 60.0, 0)
    E.update_all_fields(True)
    vals = E.latest_values(0, fields, raw=True)
    out["fields"] = {E.field_meta(v.field_id)["tag"]: (v.py(), v.status) for v in vals}
    with E.Exporter(interval_ms=1000, extras=True, profiling=True, self_metrics=True) as ex:
        ex.sweep()
        text = ex.render()
        open("gpurun_out/metrics.prom", "w").write(text)
        lat = []
        for _ in range(200):
            t = time.perf_counter(); ex.render(); lat.append((time.perf_counter() - t) * 1e3)
        lat.sort()
        out["render_ms_p50"] = lat[100]; out["render_ms_p99"] = lat[197]
    time.sleep(3)
    out["introspect_10hz_all_fields"] = E.introspect()
    E.shutdown()

if "diag" in which:
    from k8s_gpu_monitor_amd.ops import diag as D
    out["diag_device"] = D.device_info(0)
    st = {}
    for k in ["copy", "scale", "add", "triad", "read", "write"]:
        for nt in (False, True):
            try:
                st[f"{k}{'_nt' if nt else ''}"] = D.hbm_stream(k, 4 << 30, 10, nt)
            except Exception as e:
                st[f"{k}{'_nt' if nt else ''}"] = repr(e)
        print("stream", k, st[k], flush=True)
    out["stream"] = st
    out["memtest_inject"] = D.memtest(bytes=8 << 30, inject_offset=(5 << 30) + 4096, inject_xor=1 << 17)
    print("memtest inject", out["memtest_inject"]["errors"], flush=True)
    out["memtest_clean"] = D.memtest(bytes=16 << 30)
    print("memtest clean", out["memtest_clean"], flush=True)
    pk = {}
    for dt in ["bf16", "fp8", "mxfp8", "f32"]:
        try:
            pk[dt] = D.mfma_peak(dt, 1.0)
        except Exception as e:
            pk[dt] = repr(e)
        print("peak", dt, pk[dt], flush=True)
    out["mfma_peak"] = pk

json.dump(out, open("gpurun_out/probe.json", "w"), indent=1, default=str)
print("probe done")
