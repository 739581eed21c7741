"""The mapping nodes' real size: the reference scan pair (16k points each), per-call wall times of the calls a node makes per
scan -- setInputTarget, setInputSource, align (with and without fetching the aligned cloud), getFitnessScore (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import ndt
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
t, s = d["target"], d["source"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
g.setInputTarget(t); g.setInputSource(s); g.align()
def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
print("setInputTarget %.0f us | setInputSource %.0f us" % (med(lambda: g.setInputTarget(t)), med(lambda: g.setInputSource(s))))
print("align (no cloud) %.0f us | align + aligned cloud to host %.0f us | iterations %d evals %d" %
      (med(lambda: g.align()), med(lambda: g.align(n_out=len(s))), g.getFinalNumIteration(), g.stats()["n_evals"]))
print("getFitnessScore %.0f us" % med(lambda: g.getFitnessScore()))
def scan():
    g.setInputTarget(t); g.setInputSource(s); g.align(); g.getFinalTransformation()
print("per scan (target + source + align + result) %.0f us" % med(scan))
