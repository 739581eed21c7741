"""ndt_align by scan size on the headline target (set U, 1 M points, 1 m; 30 forced passes as bench.py): us per registration and
per evaluation -- for tuning the latency kernels' points per block (NDT_K2_PPB)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0); g.setNeighborhoodSearchMethod(ndt.DIRECT7)
g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
g.setInputTarget(tgt)
out = {"NDT_K2_PPB": os.environ.get("NDT_K2_PPB", "default")}
for n in [int(a) for a in sys.argv[1:]] or [16000, 30000, 50000, 65536, 80000, 100000, 131072]:
    src = clouds.source_from_target(tgt, n, seed=clouds.SEED + 1)
    g.setInputSource(src)
    for _ in range(5): g.align()
    ts = []
    for _ in range(30):
        t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0)
    st = g.stats()
    ev = st["n_evals"] + st["n_hessian_recomputes"]
    out[str(n)] = [round(float(np.median(ts)) * 1e6, 1), round(float(np.median(ts)) * 1e6 / ev, 2)]
print(json.dumps(out))
