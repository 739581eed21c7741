"""Pipelined target-build time of small clouds (the mapping nodes' sizes) for tuning the bucket plan:
  NDT_K1_SMALL_DIV=8 python tools/time_k1_small.py"""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
import torch
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
cases = [("reference pair target", d["target"].astype(np.float32)), ("16k/40m", clouds.target_surfaces(16000, extent=40.0, n_boxes=40)),
         ("60k/40m", clouds.target_surfaces(60000, extent=40.0, n_boxes=40)), ("200k/60m", clouds.target_surfaces(200000, extent=60.0, n_boxes=40))]
out = {}
for name, tgt in cases:
    n = len(tgt)
    dev = torch.from_numpy(np.c_[tgt[:, :3], np.ones(n, np.float32)]).cuda()
    g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
    for i in range(4): g.setInputTargetDevice(dev.data_ptr(), n, 16)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(20): g.setInputTargetDevice(dev.data_ptr(), n, 16)
    torch.cuda.synchronize(); out[name] = round((time.perf_counter() - t0) / 20 * 1e6, 1)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NDT_K1")}, "us_per_build": out}))
