"""configs[4]'s three levels one by one (development aid): per level, the evaluation kernels' event times of one coarse-to-fine
registration of a 2M-point scan against the 10M-point target (ndt_profile_enable(1): one launch per evaluation)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import clouds, ndt
ext = float(sys.argv[1]) if len(sys.argv) > 1 else 400.0
tgt = clouds.target_surfaces(10000000, extent=ext, n_boxes=60)
src = clouds.source_from_target(tgt, 2000000)
T = None
for res in (2.0, 1.0, 0.5):
    g = ndt.NormalDistributionsTransform(); g.setResolution(res); g.setTransformationEpsilon(0.01); g.setMaximumIterations(35); g.setStepSize(0.1)
    g.setInputTarget(tgt); g.setInputSource(src)
    g.align(T)  # warm
    g.profile(1)
    g.align(T)
    out = {"resolution": res, "iterations": g.getFinalNumIteration(), "stats": g.stats()}
    for kind, name in ((0, "with_hessian"), (1, "without"), (2, "f64_hessian")):
        n, ms = g.profile_read(kind)
        out[name] = {"launches": n, "us_each": (ms * 1e3 / n) if n else None}
    g.profile(0)
    T = g.getFinalTransformation()
    print(json.dumps(out), flush=True)
