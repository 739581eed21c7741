import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["NDT_TRACE_N1"] = "1"
import numpy as np
from toyslam_amd import ndt, clouds
rng = np.random.default_rng(3)
world = clouds.target_surfaces(240000, seed=77, extent=60.0)[:, :3].astype(np.float32)
raw = np.c_[world[rng.choice(len(world), 60000, replace=False)], np.ones(60000, np.float32)].astype(np.float32)
g = ndt.NormalDistributionsTransform()
for i in range(6):
    t0 = time.perf_counter(); c, _ = g.voxelGridFilterCloud(raw, 0.5); t1 = time.perf_counter()
    print("call %.1f us -> %d points" % ((t1 - t0) * 1e6, len(c)))
