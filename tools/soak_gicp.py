"""Soak of the GICP path (development aid): thousands of back-to-back registrations on one handle, inputs re-set now and then --
every result bit-identical, and no registration anywhere near the objective server's 20 ms patience (a stall would show there)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import gicp
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
t, s = d["target"], d["source"]
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
g = gicp.GeneralizedIterativeClosestPoint(); g.setInputTarget(t); g.setInputSource(s); g.align()
want, st = g.getFinalTransformation(), g.stats()
times, bad = [], 0
for rep in range(n):
    if rep % 50 == 49:
        g.setInputSource(s); g.setInputTarget(t)
    t0 = time.perf_counter(); g.align(want_cloud=(rep % 3 == 0)); times.append(time.perf_counter() - t0)
    if not (np.array_equal(g.getFinalTransformation(), want) and g.stats() == st):
        bad += 1
times = np.array(times) * 1e3
print("gicp soak: %d registrations, mismatches %d, align ms median %.3f p99 %.3f max %.3f" % (n, bad, np.median(times), np.percentile(times, 99), times.max()))
sys.exit(1 if bad or times.max() > 15.0 else 0)
