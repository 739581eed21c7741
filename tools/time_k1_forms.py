"""setInputTarget (cloud in HBM) over cloud shapes from uniform to heavily crowded, per K1 form:
  for k in old new auto; do NDT_K1=$k python tools/time_k1_forms.py; done
us_median_later: one build + device synchronisation; us_pipelined: six builds back to back / 6."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
import torch
cases = [("pair-like 16k/40m", 16000, 40.0, 1.0), ("30k/40m", 30000, 40.0, 1.0), ("45k/40m", 45000, 40.0, 1.0), ("60k/40m", 60000, 40.0, 1.0), ("300k/60m", 300000, 60.0, 1.0), ("1M/60m", 1000000, 60.0, 1.0),
         ("1M/100m surfaces", 1000000, 100.0, 1.0), ("1M uniform", 1000000, None, 1.0), ("2M/150m 0.5", 2000000, 150.0, 0.5), ("10M/400m 0.5", 10000000, 400.0, 0.5)]
cases = cases[:int(os.environ.get("K1_CASES", len(cases)))]  # K1_CASES=3: the small clouds only
for name, n, ext, res in cases:
    tgt = clouds.target_uniform(n) if ext is None else clouds.target_surfaces(n, extent=ext, n_boxes=40)
    dev = torch.from_numpy(np.c_[tgt, np.ones(n, np.float32)]).cuda()
    g = ndt.NormalDistributionsTransform(); g.setResolution(res)
    ts = []
    for i in range(8):
        torch.cuda.synchronize(); t0 = time.perf_counter(); g.setInputTargetDeviceRef(dev.data_ptr(), n); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(6):
        g.setInputTargetDeviceRef(dev.data_ptr(), n)
    torch.cuda.synchronize(); pipelined = (time.perf_counter() - t0) / 6
    ijk = np.floor(tgt / res).astype(np.int64); key = (ijk[:, 0] * 100003 + ijk[:, 1]) * 100003 + ijk[:, 2]
    _, c = np.unique(key, return_counts=True)
    print(json.dumps({"case": name, "K1": os.environ.get("NDT_K1", "auto"), "us_first": round(ts[0] * 1e6), "us_median_later": round(float(np.median(ts[2:])) * 1e6, 1), "us_pipelined": round(pipelined * 1e6, 1),
                      "voxels": int(len(c)), "max_cell": int(c.max()), "frac_pts_in_cells_gt48": float(c[c > 48].sum() / n)}), flush=True)
