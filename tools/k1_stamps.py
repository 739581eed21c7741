"""Phase clocks of k1_finalize (development aid): python tools/k1_stamps.py <points> <extent|0 = uniform> <resolution>"""
import os, sys
os.environ["NDT_K1_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toyslam_amd import clouds, ndt
n, ext, res = int(float(sys.argv[1])), float(sys.argv[2]), float(sys.argv[3])
tgt = clouds.target_uniform(n) if ext == 0 else clouds.target_surfaces(n, extent=ext, n_boxes=40)
dev = torch.from_numpy(np.c_[tgt, np.ones(n, np.float32)]).cuda()
g = ndt.NormalDistributionsTransform(); g.setResolution(res)
for i in range(3):
    g.setInputTargetDeviceRef(dev.data_ptr(), n)
torch.cuda.synchronize()
