"""One multi-million-point scan: ndt_align (launch path above 1.5 M points) against a lock-step batch of ONE scan (the throughput
kernels + k_reduce), per resolution -- which kernel family serves big scans better (development aid)."""
import json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
from toyslam_amd import clouds, ndt
tgt = clouds.target_surfaces(10000000, extent=400.0, n_boxes=60)
src = clouds.source_from_target(tgt, 2000000)
out = {}
for res in (0.5, 1.0, 2.0):
    g = ndt.NormalDistributionsTransform(); g.setResolution(res)
    g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
    g.setInputTarget(tgt); g.setInputSource(src)
    g.align(); torch.cuda.synchronize()
    t0 = time.perf_counter(); g.align(); t1 = time.perf_counter()
    ev = g.stats()["n_evals"]
    r = g.alignBatch(clouds=[src]); torch.cuda.synchronize()
    t2 = time.perf_counter(); r = g.alignBatch(clouds=[src]); t3 = time.perf_counter()
    evb = g.stats()["n_evals"]
    out[res] = {"align_ms": round((t1 - t0) * 1e3, 3), "evals": ev, "us_per_eval": round((t1 - t0) * 1e6 / max(ev, 1), 1),
                "batch_of_one_ms": round((t3 - t2) * 1e3, 3), "batch_evals": evb, "batch_us_per_eval": round((t3 - t2) * 1e6 / max(evb, 1), 1)}
print(json.dumps(out))
