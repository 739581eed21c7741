import sys, numpy as np
sys.path.insert(0, ".")
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
scans, Tg = [], []
for k in range(N):
    s, T = clouds.mapbuild_scan(tgt, k); scans.append(s); Tg.append(T)
def mk():
    g = ndt.NormalDistributionsTransform(); g.setMaximumIterations(28); g.setTransformationEpsilon(0.0); g.setInputTarget(tgt); return g
g = mk()
for path in (True, False):
    g.setEvaluationPath(path); g.setInputSource(scans[7]); g.align(); T = g.getFinalTransformation()
    print("single scan 7 server=%s" % path, "iters", g.getFinalNumIteration(), g.stats(), "nan", bool(np.isnan(T).any()), "tr err %.2e" % np.abs(T[:3,3]-Tg[7][:3,3]).max())
for lo, hi in ((7, 8), (0, 8), (0, 16), (0, 64), (0, 128), (0, 256), (0, N)):
    if hi > N: continue
    g = mk()
    res = g.alignBatch(scans[lo:hi])
    nan = np.isnan(res["T"]).any(axis=(1, 2))
    print("batch [%d,%d)" % (lo, hi), "nan scans", (np.nonzero(nan)[0] + lo).tolist(), "scan7 tr err %.2e" % np.abs(res["T"][7 - lo][:3,3]-Tg[7][:3,3]).max(), "iters7", res["iterations"][7 - lo], g.stats())
