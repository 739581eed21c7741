"""Agreement of the two all-f64 paths (computeHessian, calculateScore) with the oracle on the reference pair: relative errors."""
import sys, numpy as np
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from toyslam_amd import ndt, clouds
from oracle import pyoracle as po
d = np.load(__import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'tests', 'golden', 'pair_0p1.npz')); t, s = d['target'], d['source']
for res, m in ((1.0, ndt.DIRECT7), (3.0, ndt.DIRECT1), (1.0, ndt.KDTREE), (0.5, ndt.DIRECT26)):
    g = ndt.NormalDistributionsTransform(); g.setResolution(res); g.setNeighborhoodSearchMethod(m); g.setInputTarget(t); g.setInputSource(s)
    o = po.OracleNDT(resolution=res, search_method=m, num_threads=8); o.set_target(t); o.set_source(s)
    p = np.array([0.05, -0.03, 0.02, 0.004, -0.002, 0.006])
    hg, ho = g.hessian_f64(p), o.hessian_f64(p)
    cg, co = g.calculateScore(s), o.calculate_score(s)
    print(res, m, "h64 rel", np.abs(hg - ho).max() / np.abs(ho).max(), "calc rel", abs(cg - co) / abs(co), cg)
