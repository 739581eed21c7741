"""GICP row: per-phase timings (index build, covariances, one correspondence step, one objective evaluation)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from toyslam_amd import clouds, gicp
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
tgt, src = d["target"], d["source"]
if len(sys.argv) > 2:
    tgt = clouds.target_surfaces(int(sys.argv[1]))[:, :3].astype(np.float32)
    src = clouds.source_from_target(tgt, int(sys.argv[2]))[:, :3].astype(np.float32)
g = gicp.GeneralizedIterativeClosestPoint()
g.setInputTarget(tgt[:64]); g.setInputSource(src[:64]); g.align()
def med(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e3)
    return float(np.median(ts))
print("set target %.3f ms, set source %.3f ms" % (med(lambda: g.setInputTarget(tgt)), med(lambda: g.setInputSource(src))))
def cov(which):
    g.setCorrespondenceRandomness(19); g.setCorrespondenceRandomness(20)  # drops the cached covariances
    g.covariances(which)
print("covariances target %.3f ms (incl. D2H of n x 72 B), source %.3f ms" % (med(lambda: cov(0)), med(lambda: cov(1))))
g.align()
print("correspondence step %.3f ms (incl. D2H of corr + maha)" % med(lambda: g.step_correspond()))
x = np.zeros(6)
for mode in (0, 1, 2):
    print("functor mode %d: %.1f us" % (mode, 1e3 * med(lambda: g.step_functor(mode, x), 50)))
print("align %.3f ms" % med(lambda: g.align(), 7), g.stats(), g.getFinalNumIteration())
