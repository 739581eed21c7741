"""Replays one case of tools/fuzz_align.py (seed, case) and walks the product's driver twice -- fed by the GPU's evaluations
and fed by the oracle's -- printing, step by step, the request (kind, pose) and both evaluators' answers at that pose, to
tell a line search that rounding sent down another path from a wrong evaluation (development aid).
   python3 tools/probes/replay_align_case.py <seed> <case>"""
import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
from toyslam_amd import clouds, ndt
from oracle import pyoracle as po
seed, want = int(sys.argv[1]), int(sys.argv[2])
d = np.load("tests/golden/pair_0p1.npz"); t, s = d["target"], d["source"]
rng = np.random.default_rng(seed)
methods = [po.KDTREE, po.DIRECT26, po.DIRECT7, po.DIRECT1]
for case in range(want + 1):
    res = float(rng.choice([0.5, 0.8, 1.0, 1.5, 2.0, 3.0]))
    m = int(rng.choice(methods))
    kw = dict(resolution=res, search_method=m, step_size=float(rng.choice([0.05, 0.1, 0.3])), outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])),
              trans_eps=float(rng.choice([0.1, 0.01, 1e-3])), max_iter=int(rng.choice([5, 20, 35])))
    nt = int(rng.integers(2000, len(t))); ns = int(rng.integers(50, len(s)))
    tt = t[rng.choice(len(t), nt, replace=False)].copy(); ss = s[rng.choice(len(s), ns, replace=False)].copy()
    dense_t = True
    if rng.random() < 0.3:
        tt[rng.choice(nt, 5, replace=False)] = np.nan; dense_t = False
    if rng.random() < 0.3:
        ss[rng.choice(ns, 3, replace=False), int(rng.integers(0, 3))] = np.inf if rng.random() < 0.5 else np.nan
    guess = None if rng.random() < 0.5 else clouds.random_T(rng, 0.3, 2.0).astype(np.float32)
print("case", want, kw, "nt", nt, "ns", ns, "dense_t", dense_t, "guess", guess is not None)
g = ndt.NormalDistributionsTransform(); o = po.OracleNDT(num_threads=8, **kw)
g.setResolution(res); g.setNeighborhoodSearchMethod(m); g.setStepSize(kw["step_size"]); g.setOutlierRatio(kw["outlier_ratio"])
g.setTransformationEpsilon(kw["trans_eps"]); g.setMaximumIterations(kw["max_iter"])
g.setInputTarget(tt, is_dense=dense_t); o.set_target(tt, is_dense=dense_t)
g.setInputSource(ss); o.set_source(ss)
s4 = np.c_[ss, np.ones(len(ss), np.float32)]
def ev_gpu(kind, T, p):
    if kind == 2:
        return 0.0, np.zeros(6), g.hessian_f64(p)
    sc, gr, H, _ = g.eval(p, kind == 0, T)
    return sc, gr, (H if H is not None else np.zeros((6, 6)))
def ev_orc(kind, T, p):
    tc = po.transform_cloud(s4, T)
    if kind == 2:
        o.eval(p, False, tc)
        return 0.0, np.zeros(6), o.hessian_f64(p)
    sc, gr, H, _ = o.eval(p, kind == 0, tc)
    return sc, gr, (H if H is not None else np.zeros((6, 6)))
def walk(primary, other, tag):
    log = []
    def evaluator(kind, T, p):
        a = primary(kind, T, p); b = other(kind, T, p)
        rel = lambda x, y: float(np.abs(np.asarray(x, float) - np.asarray(y, float)).max() / max(np.abs(np.asarray(y, float)).max(), 1e-300))
        log.append((kind, p.copy(), a[0], rel(a[0], b[0]), rel(a[1], b[1]), rel(a[2], b[2]) if kind != 1 else 0.0))
        return a
    r = ndt.host_run_driver(evaluator, len(ss), guess, **{k: kw[k] for k in ("resolution", "step_size", "outlier_ratio", "trans_eps", "max_iter")})
    print(tag, "iterations", r["iterations"], "evals", r["n_evals"], "converged", r["converged"], "t", r["T"][:3, 3])
    return log
lg = walk(ev_gpu, ev_orc, "driver on GPU evaluations:   ")
lo = walk(ev_orc, ev_gpu, "driver on oracle evaluations:")
for k in range(max(len(lg), len(lo))):
    a = lg[k] if k < len(lg) else None; b = lo[k] if k < len(lo) else None
    same = a is not None and b is not None and a[0] == b[0] and np.array_equal(a[1], b[1])
    dp = float(np.abs(a[1] - b[1]).max()) if a is not None and b is not None else float("nan")
    print("step %2d kind %s/%s  pose diff %.3e  %s | gpu-vs-oracle at the GPU walk's pose: score %.1e grad %.1e H %.1e | score %.9g / %.9g" %
          (k, a[0] if a else "-", b[0] if b else "-", dp, "same request" if same else "DIFFERENT", a[3] if a else 0, a[4] if a else 0, a[5] if a else 0, a[2] if a else float("nan"), b[2] if b else float("nan")))
a = g.align(guess); r = o.align(guess)
print("library align: iterations", g.getFinalNumIteration(), "oracle align:", r["iterations"])
