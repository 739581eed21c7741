import sys, os, numpy as np
sys.path.insert(0, ".")
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
s, T = clouds.mapbuild_scan(tgt, 7)
g = ndt.NormalDistributionsTransform(); g.setMaximumIterations(28); g.setTransformationEpsilon(0.0); g.setInputTarget(tgt)
if sys.argv[1] == "single":
    g.setInputSource(s); g.align()
else:
    g.alignBatch([s])
