import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toyslam_amd import clouds, ndt
import torch
tgt = clouds.target_uniform(1000000)
g = ndt.NormalDistributionsTransform()
g.setInputTarget(tgt)
d = torch.from_numpy(np.c_[tgt, np.ones(len(tgt), np.float32)]).cuda(); torch.cuda.synchronize()
for _ in range(10):
    g.setInputTargetDevice(d.data_ptr(), len(tgt), 16); torch.cuda.synchronize()
