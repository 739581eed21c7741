import os, sys
os.environ["NDT_K1_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from toyslam_amd import ndt
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests", "golden", "pair_0p1.npz"))
t = d["target"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
for i in range(3): g.setInputTarget(t)
print(g.grid_counts(), t.shape, np.ptp(t, axis=0))
