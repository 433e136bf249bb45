"""What one sampler tick costs, source by source: times a direct read (one backend
round restricted to the field's source) of every non-static field on the real
device library and prints the fields grouped by cost.  Writes
gpurun_out/source_cost.json."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from k8s_gpu_monitor_amd import engine as E  # noqa: E402
from k8s_gpu_monitor_amd import smi  # noqa: E402

backend = sys.argv[1] if len(sys.argv) > 1 else "amdsmi"
smi.init(backend=backend)
rows = []
for fid in E.field_list():
    m = E.field_meta(fid)
    if m["is_static"]:
        continue
    smi.read_fields(0, [fid])
    t0 = time.perf_counter()
    c0 = time.process_time()
    n = 30
    for _ in range(n):
        v = smi.read_fields(0, [fid])
    rows.append({"field": m["tag"], "id": fid, "wall_us": round((time.perf_counter() - t0) / n * 1e6, 1),
                 "cpu_us": round((time.process_time() - c0) / n * 1e6, 1), "value": v[fid]})
rows.sort(key=lambda r: -r["wall_us"])
for r in rows:
    print(f'{r["field"]:34s} {r["wall_us"]:9.1f} us wall {r["cpu_us"]:9.1f} us cpu   {r["value"]}')
# all fields in one round: what the sampler pays per GPU per tick when everything is watched
ids = [r["id"] for r in rows]
t0, c0 = time.perf_counter(), time.process_time()
for _ in range(20):
    smi.read_fields(0, ids)
allw, allc = (time.perf_counter() - t0) / 20 * 1e6, (time.process_time() - c0) / 20 * 1e6
print(f"all {len(ids)} fields in one round: {allw:.1f} us wall, {allc:.1f} us cpu")
os.makedirs("gpurun_out", exist_ok=True)
json.dump({"backend": backend, "per_field": rows, "all_fields_wall_us": allw, "all_fields_cpu_us": allc},
          open("gpurun_out/source_cost.json", "w"), indent=1, default=str)
smi.shutdown()
