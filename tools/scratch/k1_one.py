import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toyslam_amd import clouds, ndt
import torch
n = 1000000
tgt = clouds.target_uniform(n)
dev = torch.from_numpy(np.c_[tgt, np.ones(n, np.float32)]).cuda()
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
for i in range(3): g.setInputTargetDevice(dev.data_ptr(), n, 16)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(20): g.setInputTargetDevice(dev.data_ptr(), n, 16)
torch.cuda.synchronize(); print(os.environ.get("NDT_K1_LDS_CAP"), "us per build", round((time.perf_counter() - t0) / 20 * 1e6, 1))
