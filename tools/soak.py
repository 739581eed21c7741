"""Soak test of the evaluation-server protocol (development aid): many back-to-back registrations of
several shapes; every repetition must return the bit-identical result of the first one."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
    cases = []
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(d["target"])
    g.setInputSource(d["source"])
    cases.append(("pair/DIRECT7", g))
    g = ndt.NormalDistributionsTransform()
    g.setNeighborhoodSearchMethod(ndt.KDTREE)
    g.setTransformationEpsilon(1e-9)
    g.setMaximumIterations(12)
    g.setInputTarget(d["target"])
    g.setInputSource(d["source"])
    cases.append(("pair/KDTREE/tight", g))
    tgt = clouds.target_uniform(1000000)
    g = ndt.NormalDistributionsTransform()
    g.setMaximumIterations(28)
    g.setTransformationEpsilon(0.0)
    g.setInputTarget(tgt)
    g.setInputSource(clouds.source_from_target(tgt, 100000))
    cases.append(("headline", g))
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(tgt)
    g.setInputSource(clouds.source_from_target(tgt, 300000))
    cases.append(("300k source", g))
    ref = {}
    for name, g in cases:
        g.align()
        ref[name] = (g.getFinalTransformation().copy(), g.getFinalNumIteration(), g.getTransformationProbability())
    t_end = time.time() + seconds
    n = {name: 0 for name, _ in cases}
    bad = 0
    k = 0
    t_say = time.time() + 30.0
    while time.time() < t_end:
        if time.time() > t_say:  # (a GPU box takes a run that stays silent for minutes to be hung)
            print("... registrations so far:", n, "mismatches:", bad, flush=True)
            t_say = time.time() + 30.0
        name, g = cases[k % len(cases)]
        k += 1
        reps = 200 if name.startswith("pair") else 20
        for _ in range(reps):
            g.align()
            T, it, tp = g.getFinalTransformation(), g.getFinalNumIteration(), g.getTransformationProbability()
            if not (np.array_equal(T, ref[name][0]) and it == ref[name][1] and tp == ref[name][2]):
                bad += 1
            n[name] += 1
    print("registrations:", n, "mismatches:", bad)
    assert bad == 0


if __name__ == "__main__":
    main()
