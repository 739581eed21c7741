"""ndt_voxel_grid_filter_device with the dense, the sparse and the automatically chosen voxel index (N1 prefilter)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
import torch
for n, leaf in ((100000, 0.1), (300000, 0.1), (2000000, 0.5)):
    raw = clouds.target_surfaces(n, extent=150.0, n_boxes=80)
    dev_in = torch.from_numpy(np.c_[raw, np.ones(n, np.float32)]).cuda()
    dev_out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    res = {}
    for form, name in ((1, "dense"), (2, "sparse"), (0, "auto")):
        g = ndt.NormalDistributionsTransform(); g.setVoxelIndex(form)
        try:
            m = g.voxelGridFilterDevice(dev_in.data_ptr(), n, 16, leaf, dev_out.data_ptr())
            td = []
            for _ in range(7):
                torch.cuda.synchronize(); t0 = time.perf_counter(); m = g.voxelGridFilterDevice(dev_in.data_ptr(), n, 16, leaf, dev_out.data_ptr()); td.append(time.perf_counter() - t0)
            res[name] = dict(ms=float(np.median(td)) * 1e3, voxels=int(m))
        except Exception as e:
            res[name] = repr(e)[:100]
    print(json.dumps({"points": n, "leaf": leaf, **res}))
