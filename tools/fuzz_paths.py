"""Randomised check that the persistent evaluation server and the launch-per-evaluation path return
bit-identical registrations (development aid): random sizes up to 131k source points (one point per server
thread, where both paths share the partition), search methods, parameters, guesses, NaN points.
fuzz_paths.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt  # noqa: E402


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    world = clouds.target_surfaces(600000, extent=60.0, n_boxes=40)
    bad = 0
    for case in range(n_cases):
        nt = int(rng.integers(5000, len(world)))
        ns = int(rng.integers(1, 131072))
        tt = world[rng.choice(len(world), nt, replace=False)]
        T = clouds.random_T(rng, 0.5, 3.0)
        ss = clouds.apply_T(np.linalg.inv(T), world[rng.choice(len(world), ns, replace=False)] + rng.normal(0, 0.02, (ns, 3)))
        if rng.random() < 0.3:
            ss[rng.choice(ns, min(5, ns), replace=False)] = np.nan
        guess = None if rng.random() < 0.5 else clouds.random_T(rng, 0.2, 1.0).astype(np.float32)
        res = float(rng.choice([0.5, 1.0, 2.0]))
        m = int(rng.choice([ndt.KDTREE, ndt.DIRECT26, ndt.DIRECT7, ndt.DIRECT1]))
        eps = float(rng.choice([0.1, 0.01, 1e-4, 0.0]))
        mi = int(rng.choice([3, 10, 30]))
        out = {}
        for persistent in (True, False):
            g = ndt.NormalDistributionsTransform()
            g.setResolution(res)
            g.setNeighborhoodSearchMethod(m)
            g.setTransformationEpsilon(eps)
            g.setMaximumIterations(mi)
            g.setEvaluationPath(persistent)
            g.setInputTarget(tt)
            g.setInputSource(ss)
            cloud = g.align(guess, n_out=ns)
            out[persistent] = (g.getFinalTransformation().copy(), g.getFinalNumIteration(), g.hasConverged(),
                               g.getTransformationProbability(), g.stats(), cloud)
        a, b = out[True], out[False]
        same = (np.array_equal(a[0], b[0], equal_nan=True) and a[1] == b[1] and a[2] == b[2] and
                (a[3] == b[3] or (a[3] != a[3] and b[3] != b[3])) and a[4] == b[4] and np.array_equal(a[5], b[5], equal_nan=True))
        if not same:
            bad += 1
            print("MISMATCH case", case, "nt", nt, "ns", ns, "res", res, "method", m, "eps", eps, "max_iter", mi, "it", a[1], b[1], a[4], b[4])
    print("path fuzz done, mismatches:", bad)


if __name__ == "__main__":
    main()
