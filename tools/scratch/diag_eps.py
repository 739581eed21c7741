import sys, numpy as np
sys.path.insert(0, ".")
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
scans, Tg = [], []
for k in range(512):
    s, T = clouds.mapbuild_scan(tgt, k); scans.append(s); Tg.append(T)
for eps in (1e-9, 1e-6):
    g = ndt.NormalDistributionsTransform(); g.setMaximumIterations(28); g.setTransformationEpsilon(eps); g.setInputTarget(tgt)
    res = g.alignBatch(scans)
    it = res["iterations"]
    rot = np.array([np.abs(res["T"][k][:3,:3]-Tg[k][:3,:3]).max() for k in range(512)])
    tr = np.array([np.abs(res["T"][k][:3,3]-Tg[k][:3,3]).max() for k in range(512)])
    print("eps", eps, "iters", dict(zip(*np.unique(it, return_counts=True))), "nan", int(np.isnan(res["T"]).any(axis=(1,2)).sum()), "ok", int(((rot < 2e-3) & (tr < 2e-2)).sum()), g.stats())
    single = ndt.NormalDistributionsTransform(); single.setMaximumIterations(28); single.setTransformationEpsilon(eps); single.setInputTarget(tgt)
    nd, wr, wt, nn = 0, 0, 0, 0
    for k in range(512):
        single.setInputSource(scans[k]); single.align(); T = single.getFinalTransformation()
        if np.isnan(T).any() or np.isnan(res["T"][k]).any(): nn += 1; continue
        nd += int(single.getFinalNumIteration() != it[k]); wr = max(wr, np.abs(T[:3,:3]-res["T"][k][:3,:3]).max()); wt = max(wt, np.abs(T[:3,3]-res["T"][k][:3,3]).max())
    print("   vs single: iter diffs", nd, "nan either", nn, "worst rot %.2e trans %.2e" % (wr, wt))
