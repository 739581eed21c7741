"""CPU-only randomised check of the product's host driver (Newton + More-Thuente state machine, 6x6 solve,
pose <-> matrix) against the oracle's driver: fed the oracle's own evaluations it must walk the oracle's
path (development aid).   fuzz_driver.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import pyoracle as po  # noqa: E402
from toyslam_amd import clouds, ndt  # noqa: E402


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
    t, s = d["target"], d["source"]
    exact = close = off = 0
    for case in range(n_cases):
        res = float(rng.choice([0.5, 1.0, 2.0, 3.0]))
        m = int(rng.choice([po.KDTREE, po.DIRECT26, po.DIRECT7, po.DIRECT1]))
        kw = dict(resolution=res, search_method=m, step_size=float(rng.choice([0.05, 0.1, 0.3])),
                  outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), trans_eps=float(rng.choice([0.1, 0.01, 1e-3, 1e-9, 0.0])),
                  max_iter=int(rng.choice([0, 3, 12, 35])))
        tt = t[rng.choice(len(t), int(rng.integers(1500, 8000)), replace=False)]
        ss = s[rng.choice(len(s), int(rng.integers(30, 3000)), replace=False)]
        guess = None if rng.random() < 0.4 else clouds.random_T(rng, 0.4, 3.0).astype(np.float32)
        o = po.OracleNDT(num_threads=4, **kw)
        o.set_target(tt)
        o.set_source(ss)
        ref = o.align(guess)
        s4 = np.c_[ss, np.ones(len(ss), np.float32)]

        def evaluator(kind, T, p):
            tc = po.transform_cloud(s4, T)
            if kind == 2:
                o.eval(p, False, tc)
                return 0.0, np.zeros(6), o.hessian_f64(p)
            sc, g, H, _ = o.eval(p, kind == 0, tc)
            return sc, g, H

        got = ndt.host_run_driver(evaluator, len(ss), guess=guess, resolution=res, step_size=kw["step_size"],
                                  outlier_ratio=kw["outlier_ratio"], trans_eps=kw["trans_eps"], max_iter=kw["max_iter"])
        same_counts = all(got[k] == ref[k] for k in ("converged", "iterations", "n_evals", "n_hessian_recomputes"))
        if same_counts and np.array_equal(got["T"], ref["T"], equal_nan=True):
            exact += 1
        elif same_counts and np.abs(got["T"] - ref["T"]).max() < 1e-5:
            close += 1
        else:
            off += 1
            print("OFF case", case, kw, "guess", guess is not None, "it", got["iterations"], ref["iterations"], "evals", got["n_evals"],
                  ref["n_evals"], "hess", got["n_hessian_recomputes"], ref["n_hessian_recomputes"], "dT", float(np.nanmax(np.abs(got["T"] - ref["T"]))))
    print("driver fuzz: %d bit-identical, %d same path (<1e-5), %d off" % (exact, close, off))


if __name__ == "__main__":
    main()
