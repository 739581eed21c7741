"""Experiment: configs[2] with the source partitioned live-first (points that have a valid neighbour voxel at the initial guess),
lattice order kept inside each part, against the plain lattice order.  NDT_SORT_SOURCE=0 so that the given order is used."""
import os, sys, time
os.environ["NDT_SORT_SOURCE"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from toyslam_amd import clouds, ndt
tgt = clouds.target_surfaces(10000000, extent=400.0, n_boxes=60)
src = clouds.source_from_target(tgt, 2000000, seed=clouds.SEED + 1)
res = 0.5
g = ndt.NormalDistributionsTransform(); g.setResolution(res); g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
g.setInputTarget(tgt)
# lattice order of the source (x fastest), as the library's own ordering
so = np.floor(src[:, :3] / res).astype(np.int64); so -= so.min(0); d2 = so.max(0) + 1
order = np.argsort((so[:, 2] * d2[1] + so[:, 1]) * d2[0] + so[:, 0], kind="stable")
A = src[order]
# liveness at the identity guess from the grid's own valid voxels (dump: idx of leaves with n >= 6 and valid)
G = g.grid()
mb, db = G["min_b"].astype(np.int64), G["div_b"].astype(np.int64)
valid = np.zeros(int(db.prod()), bool)
valid[G["idx"][G["n"] >= 6]] = True
ijk = np.floor(A[:, :3] / np.float32(res)).astype(np.int64) - mb
live = np.zeros(len(A), bool)
for dx, dy, dz in ((0,0,0),(1,0,0),(-1,0,0),(0,1,0),(0,-1,0),(0,0,1),(0,0,-1)):
    q = ijk + [dx, dy, dz]
    ok = ((q >= 0) & (q < db)).all(1)
    lin = (q[:, 0] + q[:, 1] * db[0] + q[:, 2] * db[0] * db[1])
    lin[~ok] = 0
    live |= ok & valid[lin]
print("live fraction", live.mean())
B = np.concatenate([A[live], A[~live]])
def run(x, tag):
    g.setInputSource(x); g.align()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0)
    print(tag, "ms", round(min(ts) * 1e3, 3), "reg/s", round(1 / min(ts), 1), "iters", g.getFinalNumIteration(), "evals", g.stats()["n_evals"], g.getFinalTransformation()[:3, 3])
run(A, "lattice order      ")
run(B, "live-first partition")
run(A, "lattice order      ")
run(B, "live-first partition")
