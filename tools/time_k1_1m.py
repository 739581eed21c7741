"""Pipelined target-build time of the headline target (1M points, set U) and of a clustered 1M-point scene: tuning aid."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
import torch
out = {}
for name, tgt in (("1M uniform", clouds.target_uniform(1000000)), ("1M surfaces 100 m", clouds.target_surfaces(1000000, extent=100.0, n_boxes=40))):
    n = len(tgt)
    dev = torch.from_numpy(np.c_[tgt, np.ones(n, np.float32)]).cuda()
    g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
    for i in range(4): g.setInputTargetDevice(dev.data_ptr(), n, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): g.setInputTargetDevice(dev.data_ptr(), n, 16)
    torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NDT_K1")}, "us_per_build": out}))
