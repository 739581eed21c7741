"""Median wall time of ndt_align on the headline workload (development aid; NDT_TIMING=1 adds the library's own breakdown)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000); src = clouds.source_from_target(tgt, 100000)
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0); g.setMaximumIterations(28); g.setTransformationEpsilon(0.0)
g.setInputTarget(tgt); g.setInputSource(src)
for _ in range(6): g.align()
ts=[]
for _ in range(10):
    t0=time.perf_counter(); g.align(); ts.append(time.perf_counter()-t0)
print("align median %.1f us"%(np.median(ts)*1e6))
