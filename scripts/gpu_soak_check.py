"""What the soak's own checks cost: sustained GEMM rate against the share of results that get a checksum.
Output: JSON lines (dtype, shape, check_every, tflops, tflops_best)."""
import json
import sys

from k8s_gpu_monitor_amd.ops import diag

out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
for dtype in ("bf16", "mxfp8"):
    for n in (4096, 8192):
        for every in (1, 2, 4, 8, 32):
            g = diag.gemm_stress(dtype, n, n, n, seconds=2.0, check_every=every)
            print(json.dumps(dict(dtype=dtype, n=n, check_every=every, tflops=round(g["tflops"], 1),
                                  best=round(g["tflops_best"], 1), launches=g["launches"], sdc=g["sdc_mismatches"])), file=out, flush=True)
        g = diag.gemm_stress(dtype, n, n, n, seconds=2.0, verify=False)
        print(json.dumps(dict(dtype=dtype, n=n, check_every=0, tflops=round(g["tflops"], 1), best=round(g["tflops_best"], 1))), file=out, flush=True)
