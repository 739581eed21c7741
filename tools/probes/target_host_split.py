"""setInputTarget / setInputSource at the reference pair: the host's part of the call (the GPU idle when it starts) against the
call + the GPU's completion, and the back-to-back figure tools/time_pair.py reports (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from toyslam_amd import ndt
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
d = np.load(os.path.join(root, "tests", "golden", "pair_0p1.npz"))
t, s = np.ascontiguousarray(d["target"]), np.ascontiguousarray(d["source"])
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
g.setInputTarget(t); g.setInputSource(s); g.align()
def split(f, n=60):
    a, b = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        a.append((t1 - t0) * 1e6); b.append((t2 - t0) * 1e6)
    return float(np.median(a)), float(np.median(b))
def b2b(f, n=60):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
out = {}
out["target_call_us"], out["target_call_and_gpu_us"] = split(lambda: g.setInputTarget(t))
out["source_call_us"], out["source_call_and_gpu_us"] = split(lambda: g.setInputSource(s))
out["empty_sync_us"] = split(lambda: None)[1]
out["target_back_to_back_us"] = b2b(lambda: g.setInputTarget(t))
out["source_back_to_back_us"] = b2b(lambda: g.setInputSource(s))
def scan():
    g.setInputTarget(t); g.setInputSource(s); g.align(); g.getFinalTransformation()
out["per_scan_us"] = b2b(scan)
def scan_parts():
    t0 = time.perf_counter(); g.setInputTarget(t); t1 = time.perf_counter(); g.setInputSource(s); t2 = time.perf_counter(); g.align(); t3 = time.perf_counter()
    g.getFinalTransformation(); t4 = time.perf_counter()
    return [(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6]
parts = np.median(np.array([scan_parts() for _ in range(60)]), axis=0)
out["in_loop_target_source_align_result_us"] = [float(x) for x in parts]
print(json.dumps(out))
