"""NDT_TIMING=1 breakdown of align on the reference pair (run with NDT_TIMING=1; stderr carries the library's lines)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import ndt
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
t, s = d["target"], d["source"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
for i in range(12):
    g.setInputTarget(t); g.setInputSource(s)
    t0 = time.perf_counter(); g.align(); t1 = time.perf_counter()
    sys.stderr.write("align wall %.1f us\n" % ((t1 - t0) * 1e6))
