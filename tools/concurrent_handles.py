"""Throughput of several host threads registering scans on handles of their own against one shared target (one GPU):
  python tools/concurrent_handles.py [threads] [registrations per thread]"""
import json, os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt

n_threads = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n_reg = int(sys.argv[2]) if len(sys.argv) > 2 else 100
tgt = clouds.target_uniform(1000000)
base = ndt.NormalDistributionsTransform(); base.setMaximumIterations(28); base.setTransformationEpsilon(1e-9); base.setInputTarget(tgt)
handles = []
for k in range(n_threads):
    g = base.clone() if hasattr(base, "clone") else None
    if g is None:
        g = ndt.NormalDistributionsTransform(); g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9); g.setInputTarget(tgt)
    g.setInputSource(clouds.source_from_target(tgt, 100000, seed=clouds.SEED + 1 + k))
    g.align()
    handles.append(g)
def work(g):
    for _ in range(n_reg):
        g.align()
out = {}
for nt in sorted({1, 2, n_threads}):
    th = [threading.Thread(target=work, args=(handles[k],)) for k in range(nt)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    out[str(nt)] = round(nt * n_reg / dt, 1)
print(json.dumps({"env": {k: v for k, v in os.environ.items() if k.startswith("NDT_")}, "registrations_per_s_by_threads": out}))
