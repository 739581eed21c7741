"""N1 (pcl::VoxelGrid scan prefilter, ndt_voxel_grid_filter) at the mapping nodes' size: a raw scan of 60 k points over a 60 m scene,
0.5 m leaf -- host buffer in, host buffer out, and device in / device out; per call, median of 30 (development aid)."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from toyslam_amd import clouds, ndt
n_raw = int(sys.argv[1]) if len(sys.argv) > 1 else 60000
leaf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
rng = np.random.default_rng(3)
world = clouds.target_surfaces(4 * n_raw, seed=77, extent=60.0)[:, :3].astype(np.float32)
scan = (world[rng.choice(len(world), n_raw, replace=False)] + rng.normal(0, 0.01, (n_raw, 3))).astype(np.float32)
g = ndt.NormalDistributionsTransform()
def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
out0 = g.voxelGridFilter(scan, leaf)
dev = torch.from_numpy(np.c_[scan, np.ones(n_raw, np.float32)]).cuda()
dout = torch.empty((n_raw, 4), dtype=torch.float32, device="cuda")
g.voxelGridFilterDevice(dev.data_ptr(), n_raw, 16, leaf, dout.data_ptr())
print(json.dumps({"points": n_raw, "leaf": leaf, "kept": int(len(out0)),
                  "host_to_host_us": med(lambda: g.voxelGridFilter(scan, leaf)),
                  "device_to_device_us": med(lambda: g.voxelGridFilterDevice(dev.data_ptr(), n_raw, 16, leaf, dout.data_ptr()))}))
