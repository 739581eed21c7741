"""Randomised check of ndt_get_fitness_score (exact nearest neighbour over the voxel grid) against a brute
force search with the same f32 arithmetic (development aid).   fuzz_fitness.py [seed] [cases]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt  # noqa: E402


def brute(target, moved, max_range):
    f = np.float32
    best = np.full(len(moved), np.inf, dtype=f)
    fin = np.isfinite(target).all(axis=1)
    tg = target[fin].astype(f)
    for a in range(0, len(tg), 4096):
        t = tg[a:a + 4096]
        dx = moved[:, None, 0] - t[None, :, 0]
        dy = moved[:, None, 1] - t[None, :, 1]
        dz = moved[:, None, 2] - t[None, :, 2]
        d2 = (dx * dx + dy * dy).astype(f) + (dz * dz).astype(f)
        best = np.minimum(best, d2.min(axis=1))
    ok = np.isfinite(moved).all(axis=1) & (best.astype(np.float64) <= max_range)
    return float(best[ok].astype(np.float64).sum() / ok.sum()) if ok.any() else np.finfo(np.float64).max


def main():
    rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    bad = 0
    for case in range(n_cases):
        nt = int(rng.integers(10, 20000))
        ns = int(rng.integers(1, 1500))
        extent = float(rng.choice([2.0, 20.0, 200.0, 2000.0]))
        kind = rng.integers(0, 3)
        if kind == 0:
            tgt = rng.uniform(-extent, extent, (nt, 3))
        elif kind == 1:  # clusters
            c = rng.uniform(-extent, extent, (max(1, nt // 500), 3))
            tgt = c[rng.integers(0, len(c), nt)] + rng.normal(0, extent * 0.01, (nt, 3))
        else:  # thin sheet
            tgt = rng.uniform(-extent, extent, (nt, 3)) * [1, 1, 0.001]
        tgt = (tgt + rng.uniform(-1000, 1000, 3) * (rng.random() < 0.3)).astype(np.float32)
        dense = True
        if rng.random() < 0.2:
            tgt[rng.choice(nt, min(3, nt), replace=False)] = np.nan
            dense = False
        src = (tgt[rng.integers(0, nt, ns)] + rng.normal(0, extent * float(rng.choice([1e-4, 0.05, 1.0, 10.0])), (ns, 3))).astype(np.float32)
        src = np.nan_to_num(src, nan=0.0)
        if rng.random() < 0.2:
            src[rng.integers(0, ns)] = np.nan
        res = float(rng.choice([0.3, 1.0, 2.5, 10.0])) * max(extent / 50.0, 0.05)
        g = ndt.NormalDistributionsTransform()
        g.setResolution(res)
        g.setMaximumIterations(0)
        try:
            g.setInputTarget(tgt, is_dense=dense)
        except ndt.NdtError:
            continue  # index-space overflow: the reference refuses too
        g.setInputSource(src)
        guess = None if rng.random() < 0.5 else clouds.random_T(rng, extent * 0.05, 5.0).astype(np.float32)
        moved = g.align(guess, n_out=ns)[:, :3]
        for mr in (np.finfo(np.float64).max, float(rng.choice([1e-3, 1.0, 100.0])) * extent * 0.01):
            got = g.getFitnessScore(mr)
            ref = brute(tgt, moved, mr)
            if not (got == ref or abs(got - ref) <= 1e-12 * abs(ref)):
                bad += 1
                print("MISMATCH case", case, "nt", nt, "ns", ns, "extent", extent, "kind", int(kind), "res", res, "max_range", mr, got, ref)
    print("fitness fuzz done, mismatches:", bad)


if __name__ == "__main__":
    main()
