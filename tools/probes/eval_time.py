"""configs[2] scene: one with-Hessian evaluation at the identity pose (the launch path's k_derivatives_fused), timed per call,
for A/B runs of library builds (development aid):  python3 tools/probes/eval_time.py [lib names ...]"""
import os, sys, time, shutil
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
names = sys.argv[1:]
if names and not os.environ.get("EVAL_TIME_CHILD"):
    import subprocess
    for v in names + names:
        shutil.copy(os.path.join(root, "toyslam_amd", "libndt_%s.so" % v), os.path.join(root, "toyslam_amd", "libndt_mi355.so"))
        subprocess.run([sys.executable, os.path.abspath(__file__), v], env=dict(os.environ, EVAL_TIME_CHILD="1"), check=True)
    shutil.copy(os.path.join(root, "toyslam_amd", "libndt_%s.so" % names[0]), os.path.join(root, "toyslam_amd", "libndt_mi355.so"))
    sys.exit(0)
import numpy as np
from toyslam_amd import clouds, ndt
tgt = clouds.target_surfaces(10000000, extent=400.0, n_boxes=60)
src = clouds.source_from_target(tgt, 2000000, seed=clouds.SEED + 1)
g = ndt.NormalDistributionsTransform(); g.setResolution(0.5)
g.setInputTarget(tgt); g.setInputSource(src)
p = np.zeros(6)
for _ in range(5): g.eval(p)
ts = []
for _ in range(40):
    t0 = time.perf_counter(); r = g.eval(p); ts.append((time.perf_counter() - t0) * 1e6)
ts.sort()
t0 = time.perf_counter(); g.align(); t1 = time.perf_counter(); g.align(); t2 = time.perf_counter()
import hashlib
h = hashlib.sha1(np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel() for x in r])).tobytes()).hexdigest()[:12]
r2 = g.eval(p, compute_hessian=False)
h2 = hashlib.sha1(np.ascontiguousarray(np.concatenate([np.atleast_1d(np.asarray(x, dtype=np.float64)).ravel() for x in r2])).tobytes()).hexdigest()[:12]
print(names[0] if names else os.environ.get("NDT_K2_AHEAD", "lib"), "eval us: min %.1f median %.1f | align ms %.3f evals %d | score %.17g | result hashes %s %s" % (ts[0], ts[len(ts) // 2], (t2 - t1) * 1e3, g.stats()["n_evals"], r[0], h, h2), flush=True)
