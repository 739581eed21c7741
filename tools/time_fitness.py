"""getFitnessScore by how far the source lies from the target (development aid): the reference pair, the target against itself,
and the pair with the far queries (NN beyond 1 m) removed."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scipy.spatial import cKDTree
from toyslam_amd import ndt
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
t, s = d["target"], d["source"]
dist, _ = cKDTree(t).query(s)
def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return round(float(np.median(ts)), 1)
out = {}
for name, src in (("pair", s), ("target against itself", t), ("pair, NN within 1 m only", s[dist < 1.0]), ("pair, NN within 0.3 m only", s[dist < 0.3]),
                  ("pair, NN beyond 1 m only", s[dist >= 1.0])):
    g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
    g.setInputTarget(t); g.setInputSource(np.ascontiguousarray(src)); g.setMaximumIterations(0); g.align()
    g.getFitnessScore()
    out[name] = {"points": len(src), "us": med(lambda: g.getFitnessScore())}
print(json.dumps(out))
