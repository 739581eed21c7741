"""setInputSource / setInputTarget from a HOST buffer by cloud size (run once per NDT_HOST_STAGE_MAX setting): the call's
own wall time and the time until the device has finished with it."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
import torch
out = {}
tgt = clouds.target_uniform(1000000)
for n in (4096, 16384, 32768, 65536):
    pts = np.ascontiguousarray(tgt[:n])
    g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
    for _ in range(5):
        g.setInputSource(pts); g.setInputTarget(pts)
    torch.cuda.synchronize()
    def med(f, reps=40):
        call, done = [], []
        for _ in range(reps):
            t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
            call.append((t1 - t0) * 1e6); done.append((t2 - t0) * 1e6)
        return round(float(np.median(call)), 1), round(float(np.median(done)), 1)
    out[n] = {"source_call_done_us": med(lambda: g.setInputSource(pts)), "target_call_done_us": med(lambda: g.setInputTarget(pts))}
print(json.dumps({"NDT_HOST_STAGE_MAX": os.environ.get("NDT_HOST_STAGE_MAX"), "by_points": out}))
