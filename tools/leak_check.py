"""Device-memory leak check (development aid): 150 create / set inputs / align / fitness / map update / destroy cycles of an NDT
and a GICP handle with growing cloud sizes; prints the free device memory before and after (the difference is the pool's cache)."""
import sys, os, ctypes as C; sys.path.insert(0, os.getcwd())
import numpy as np
from toyslam_amd import gicp, ndt, clouds, _lib
L = _lib.lib()
import re
linked = [m.group(1) for m in re.finditer(r"(/\S*libamdhip64\.so[.\d]*)", open("/proc/self/maps").read()) if "/torch/" not in m.group(1)]
hip = C.CDLL(linked[0])
def free_mb():
    f, t = C.c_size_t(0), C.c_size_t(0); hip.hipMemGetInfo(C.byref(f), C.byref(t)); return f.value / 1e6
d = np.load("tests/golden/pair_0p1.npz"); t, s = d["target"], d["source"]
g = gicp.GeneralizedIterativeClosestPoint(); g.setInputTarget(t); g.setInputSource(s); g.align(); del g
n = ndt.NormalDistributionsTransform(); n.setInputTarget(t); n.setInputSource(s); n.align(n_out=len(s)); del n
m0 = free_mb()
for rep in range(150):
    g = gicp.GeneralizedIterativeClosestPoint(); g.setInputTarget(t[: 8000 + 50 * rep]); g.setInputSource(s[: 9000 + 40 * rep]); g.align(want_cloud=True); g.getFitnessScore(); del g
    n = ndt.NormalDistributionsTransform(); n.setInputTarget(t[: 8000 + 50 * rep]); n.setInputSource(s[: 9000 + 40 * rep]); n.align(n_out=9000 + 40 * rep); n.getFitnessScore(); n.mapUpdate(s, None, 0.5); del n
m1 = free_mb()
print("free device memory before %.1f MB, after 150 create/use/destroy cycles %.1f MB, delta %.1f MB" % (m0, m1, m0 - m1))
import resource
print("host max RSS %.0f MB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024))
