"""N1 measurement: pcl::VoxelGrid-style centroid down-sample of a raw scan, GPU vs the CPU oracle."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
from oracle import pyoracle as po
import torch

def main():
    n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 2000000
    leaf = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    raw = clouds.target_surfaces(n, extent=150.0, n_boxes=80)
    g = ndt.NormalDistributionsTransform()
    g.voxelGridFilter(raw, leaf)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); out = g.voxelGridFilter(raw, leaf); ts.append(time.perf_counter() - t0)
    dev_in = torch.from_numpy(np.c_[raw, np.ones(n, np.float32)]).cuda()
    dev_out = torch.empty((n, 4), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    td = []
    for _ in range(5):
        t0 = time.perf_counter(); m = g.voxelGridFilterDevice(dev_in.data_ptr(), n, 16, leaf, dev_out.data_ptr()); td.append(time.perf_counter() - t0)
    t0 = time.perf_counter(); ref, _ = po.voxel_grid_filter(raw, leaf); tc = time.perf_counter() - t0
    print(json.dumps({"op": "voxel_grid_filter", "points": n, "leaf_m": leaf, "voxels": int(len(out)),
                      "gpu_ms_host_buffers": float(np.median(ts)) * 1e3, "gpu_ms_device_resident": float(np.median(td)) * 1e3,
                      "cpu_oracle_ms_1thread": tc * 1e3, "identical_to_oracle": bool(np.array_equal(out, ref)),
                      "algorithmic_GBs_device_resident": (n * 16 + len(out) * 16) / np.median(td) / 1e9}))

if __name__ == "__main__":
    main()
