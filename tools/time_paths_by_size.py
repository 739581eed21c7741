"""Registration time by source size through the evaluation server and through the launch path (one kernel per
evaluation), 10M-point target at 0.5 m: where does the launch path overtake the server?
  NDT_K2_MAX_BLOCKS=512 python tools/time_paths_by_size.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt

tgt = clouds.target_surfaces(10000000, extent=400.0, n_boxes=60)
g = ndt.NormalDistributionsTransform(); g.setResolution(0.5); g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
g.setInputTarget(tgt)
out = {}
for n in (200000, 500000, 1000000, 2000000):
    src = clouds.source_from_target(tgt, n)
    g.setInputSource(src)
    row = {}
    for name, path in (("server", True), ("launch", False)):
        g.setEvaluationPath(path)
        for _ in range(2): g.align()
        t0 = time.perf_counter()
        for _ in range(8): g.align()
        row[name + "_ms"] = round((time.perf_counter() - t0) / 8 * 1e3, 3)
    row["evals"] = g.stats()["n_evals"]
    out[str(n)] = row
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NDT_")}, "by_source_points": out}))
