#!/usr/bin/env python3
"""Generates tests/golden/* (run here, where /root/reference exists; the GPU box
only ever reads the committed files).

Inputs  : the reference's bundled scan pair ndt_omp/data/251370668.pcd (target)
          and 251371071.pcd (source), voxel-downsampled at 0.1 m exactly as
          ndt_omp/apps/align.cpp:60-69 does before registering (derived DATA,
          xyz only, float32).
Outputs : what the CPU oracle (oracle/ndt_oracle.cpp) computes on them -- grid
          records, derivative evaluations, full alignments.  These are
          regression pins of the oracle; the only reference-side known answers
          (README fitness values) are checked in tests/test_oracle_readme.py.
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import pyoracle as po  # noqa: E402
from toyslam_amd import clouds  # noqa: E402

REF_DATA = "/root/reference/ndt_omp/data"
OUT = os.path.join(ROOT, "tests", "golden")


def main():
    os.makedirs(OUT, exist_ok=True)
    tgt, _ = clouds.read_pcd(os.path.join(REF_DATA, "251370668.pcd"))
    src, _ = clouds.read_pcd(os.path.join(REF_DATA, "251371071.pcd"))
    t = clouds.voxel_downsample(tgt, 0.1)
    s = clouds.voxel_downsample(src, 0.1)
    np.savez_compressed(os.path.join(OUT, "pair_0p1.npz"), target=t, source=s)

    gold = {}
    # ---- grid at 1.0 m --------------------------------------------------
    o = po.OracleNDT(resolution=1.0, search_method=po.DIRECT7, num_threads=1)
    o.set_target(t)
    o.set_source(s)
    g = o.grid()
    np.savez_compressed(os.path.join(OUT, "grid_1p0.npz"), **g)
    gold["grid_1p0"] = dict(n_leaves=int(len(g["idx"])), n_ge6=int((g["n"] >= 6).sum()),
                            n_rejected=int((g["n"] < 0).sum()), div_b=g["div_b"].tolist(),
                            min_b=g["min_b"].tolist())
    gold["gauss_1p0_0p55"] = o.gauss().tolist()

    # ---- derivative evaluations -------------------------------------------
    evals = {}
    poses = {"zero": [0, 0, 0, 0, 0, 0], "small": [0.4, 0.1, -0.02, 0.004, -0.001, -0.01],
             "large": [-0.7, 0.9, 0.15, 0.05, -0.08, 0.3]}
    for method, mname in ((po.DIRECT7, "DIRECT7"), (po.DIRECT1, "DIRECT1"), (po.DIRECT26, "DIRECT26"), (po.KDTREE, "KDTREE")):
        o.set(search_method=method)
        for pname, p in poses.items():
            score, grad, H, nn = o.eval(p, True)
            evals["%s/%s" % (mname, pname)] = dict(p=p, score=score, g=grad.tolist(), H=H.tolist(), mean_neighbors=nn,
                                                   H64=o.hessian_f64(p).tolist())
    gold["evals"] = evals
    o.set(search_method=po.DIRECT7)
    moved = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)], po.pose_to_matrix(poses["small"]))
    gold["calculate_score_small_DIRECT7"] = o.calculate_score(moved[:, :3])

    # ---- full alignments --------------------------------------------------
    aligns = {}
    guess = clouds.make_T([0.3, 0.1, -0.05], np.deg2rad([-0.4, 0.3, 0.8])).astype(np.float32)
    neg_roll = clouds.make_T([0.2, 0.0, 0.0], np.deg2rad([-1.0, 0.5, -0.5])).astype(np.float32)
    cases = [("DIRECT7/default", po.DIRECT7, None, 0.1, 35, 0.1), ("DIRECT1/default", po.DIRECT1, None, 0.1, 35, 0.1),
             ("DIRECT7/node_params", po.DIRECT7, None, 0.01, 64, 0.1),           # ndt_omp_mapping_node.cpp:38-47
             ("DIRECT7/guess", po.DIRECT7, guess, 0.01, 64, 0.1),                # ndt_rosbag_mapping_node.cpp:130
             ("DIRECT7/guess_neg_roll", po.DIRECT7, neg_roll, 0.01, 64, 0.1),    # eulerAngles [0,pi] branch
             ("DIRECT7/tight", po.DIRECT7, None, 1e-9, 28, 0.1),                 # line search iterates, f64 Hessian
             ("DIRECT26/default", po.DIRECT26, None, 0.1, 35, 0.1),
             ("KDTREE/default", po.KDTREE, None, 0.1, 35, 0.1),                  # apps/align.cpp:88-93, README.md:18-21
             ("KDTREE/node_params", po.KDTREE, None, 0.01, 64, 0.1)]
    for name, method, gs, eps, mi, step in cases:
        o.set(search_method=method, trans_eps=eps, max_iter=mi, step_size=step)
        r = o.align(gs)
        aligns[name] = dict(method=int(method), guess=None if gs is None else gs.tolist(), trans_eps=eps, max_iter=mi,
                            step_size=step, T=r["T"].tolist(), converged=r["converged"], iterations=r["iterations"],
                            trans_probability=r["trans_probability"], n_evals=r["n_evals"],
                            n_hessian_recomputes=r["n_hessian_recomputes"])
    gold["aligns"] = aligns
    gold["readme_fitness"] = {"DIRECT7": 0.214205, "DIRECT1": 0.208511, "KDTREE": 0.213937}  # ndt_omp/README.md:13-46

    with open(os.path.join(OUT, "oracle_golden.json"), "w") as f:
        json.dump(gold, f, indent=1)
    print("wrote", OUT, {k: os.path.getsize(os.path.join(OUT, k)) for k in os.listdir(OUT)})


if __name__ == "__main__":
    main()
