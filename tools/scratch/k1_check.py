import sys, time, os, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toyslam_amd import clouds, ndt
import torch
for name, tgt in (("U 1M", clouds.target_uniform(1000000)), ("S 1M", clouds.target_surfaces(1000000)), ("pair", np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests/golden/pair_0p1.npz"))["target"])):
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(tgt)
    d = torch.from_numpy(np.c_[tgt, np.ones(len(tgt), np.float32)]).cuda(); torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); g.setInputTargetDevice(d.data_ptr(), len(tgt), 16); g.grid_counts(); ts.append(time.perf_counter() - t0)
    print(name, os.environ.get("NDT_K1", "new"), "build+counts median %.1f us min %.1f" % (np.median(ts) * 1e6, min(ts) * 1e6), g.grid_counts())
