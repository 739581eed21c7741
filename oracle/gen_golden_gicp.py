"""Generates tests/golden/gicp_golden.json -- regression pins of the GICP oracle (oracle/gicp_oracle.cpp) on the committed
scan pair.  TEST INFRASTRUCTURE ONLY.  The reference has no GICP known answers (its README tabulates NDT only), so these
vectors pin the RESTATEMENT against accidental change; they are not outputs of the reference binary.
    python oracle/gen_golden_gicp.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from toyslam_amd import clouds  # noqa: E402


def main():
    d = np.load(os.path.join(ROOT, "tests", "golden", "pair_0p1.npz"))
    t, s = d["target"], d["source"]
    gold = {}
    cov = po.gicp_covariances(t, 20, 1e-3)
    idx, d2 = po.gicp_knn(t, t[:64], 20)
    gold["target_cov_sum"] = cov.sum(axis=0).tolist()
    gold["target_cov_first"] = cov[0].tolist()
    gold["knn_first64_idx_sum"] = int(idx.astype(np.int64).sum())
    gold["knn_first64_d2_sum"] = float(d2.astype(np.float64).sum())
    o = po.OracleGICP()
    o.setInputTarget(t)
    o.setInputSource(s)
    o.prepare()
    m, ci, maha = o.correspond(np.eye(4))
    x = [0.05, -0.02, 0.01, 0.003, -0.002, 0.01]
    f0, _ = o.functor(0, x)
    f2, g2 = o.functor(2, x)
    gold["step"] = dict(correspondences=int(m), corr_idx_sum=int(ci.astype(np.int64).sum()), maha_sum=float(maha.astype(np.float64).sum()),
                        x=x, f_operator=f0, f_fdf=f2, g_fdf=g2.tolist())
    aligns = {}
    guess = clouds.make_T([0.3, 0.1, -0.05], np.deg2rad([-0.4, 0.3, 0.8])).astype(np.float32)
    for name, kw, gs in (("default", {}, None), ("guess", {}, guess), ("k10_gate1", dict(k=10, corr_dist_threshold=1.0), None),
                         ("caps", dict(max_iterations=2, max_inner_iterations=4), None)):
        og = po.OracleGICP(**kw)
        og.setInputTarget(t)
        og.setInputSource(s)
        r = og.align(gs)
        aligns[name] = dict(params=kw, guess=None if gs is None else gs.tolist(), T=r["T"].tolist(), converged=r["converged"],
                            iterations=r["iterations"], n_f=r["n_f"], n_df=r["n_df"], n_fdf=r["n_fdf"], correspondences=r["correspondences"])
    gold["aligns"] = aligns
    out = os.path.join(ROOT, "tests", "golden", "gicp_golden.json")
    with open(out, "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", out, os.path.getsize(out))


if __name__ == "__main__":
    main()
