import sys, numpy as np
sys.path.insert(0, ".")
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
scans, Tg = [], []
for k in range(512):
    s, T = clouds.mapbuild_scan(tgt, k); scans.append(s); Tg.append(T)
def err(res):
    rot = np.array([np.abs(res["T"][k][:3,:3]-Tg[k][:3,:3]).max() for k in range(512)])
    tr = np.array([np.abs(res["T"][k][:3,3]-Tg[k][:3,3]).max() for k in range(512)])
    return rot, tr
for mi, eps in ((28, 0.0), (35, 0.1), (35, 0.01)):
    g = ndt.NormalDistributionsTransform(); g.setMaximumIterations(mi); g.setTransformationEpsilon(eps); g.setInputTarget(tgt)
    res = g.alignBatch(scans)
    rot, tr = err(res)
    it = res["iterations"]
    print("max_iter", mi, "eps", eps, "iters hist", np.bincount(it)[np.bincount(it) > 0], np.nonzero(np.bincount(it))[0], "converged", int(res["converged"].sum()))
    print("   rot err pct 50/90/99/max %.2e %.2e %.2e %.2e   trans %.2e %.2e %.2e %.2e" % (*np.percentile(rot, [50, 90, 99, 100]), *np.percentile(tr, [50, 90, 99, 100])))
    print("   n(rot<2e-3 & tr<2e-2) =", int(((rot < 2e-3) & (tr < 2e-2)).sum()), " nan T:", int(np.isnan(res["T"]).any(axis=(1,2)).sum()))
    bad = np.nonzero(~((rot < 2e-3) & (tr < 2e-2)))[0][:8]
    for k in bad: print("    scan", k, "iters", it[k], "rot %.2e tr %.2e" % (rot[k], tr[k]), "T_gt t", np.round(Tg[k][:3,3], 3))
