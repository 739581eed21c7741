"""GICP row: GPU vs oracle on a synthetic scene and (if present) the bundled pair; prints timings (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401  (initialises the GPU runtime before the library)
from toyslam_amd import clouds, gicp
from oracle import pyoracle as po

n_t, n_s = int(sys.argv[1]) if len(sys.argv) > 1 else 20000, int(sys.argv[2]) if len(sys.argv) > 2 else 8000
tgt = clouds.target_surfaces(n_t)[:, :3].astype(np.float32)
src = clouds.source_from_target(tgt, n_s)[:, :3].astype(np.float32)
g = gicp.GeneralizedIterativeClosestPoint()
t0 = time.time(); g.setInputTarget(tgt); g.setInputSource(src); print("set inputs %.2f ms" % ((time.time() - t0) * 1e3))
o = po.OracleGICP(); o.setInputTarget(tgt); o.setInputSource(src)

t0 = time.time(); cov, idx, d2 = g.covariances(0, neighbors=True); print("gpu covariances+nn %.2f ms" % ((time.time() - t0) * 1e3))
oi, od = po.gicp_knn(tgt, tgt, 20)
print("knn idx equal:", np.array_equal(idx, oi), " d2 equal:", np.array_equal(d2, od))
ocov = po.gicp_covariances(tgt, 20, 1e-3)
print("cov max abs diff:", np.abs(cov - ocov).max())

o.prepare()
m_o, idx_o, maha_o = o.correspond(np.eye(4))
m_g, idx_g, maha_g = g.step_correspond()
print("corr", m_o, m_g, "idx equal:", np.array_equal(idx_o, idx_g), " maha max rel diff:",
      (np.abs(maha_o - maha_g)[idx_o >= 0] / (np.abs(maha_o)[idx_o >= 0].max())).max())
x = np.array([0.05, -0.02, 0.01, 0.003, -0.002, 0.01])
for mode in (0, 1, 2):
    fo, go = o.functor(mode, x); fg, gg = g.step_functor(mode, x)
    print("mode", mode, "f", fo, fg, "rel", abs(fo - fg) / max(abs(fo), 1e-300), " g maxrel", np.abs(go - gg).max() / max(np.abs(go).max(), 1e-300))

t0 = time.time(); ro = o.align(); t_o = time.time() - t0
g.align()
ts = []
for _ in range(5):
    t0 = time.time(); g.align(); ts.append(time.time() - t0)
T = g.getFinalTransformation()
print("oracle %.1f ms iters %d f/df/fdf %d/%d/%d" % (t_o * 1e3, ro["iterations"], ro["n_f"], ro["n_df"], ro["n_fdf"]))
print("gpu    %.2f ms iters %d" % (np.median(ts) * 1e3, g.getFinalNumIteration()), g.stats(), "converged", g.hasConverged())
print("T diff max", np.abs(T - ro["T"]).max(), " vs gt", np.abs(T - clouds.T_GT_DEFAULT).max())
print("fitness", g.getFitnessScore())
