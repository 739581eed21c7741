"""Quick timing probe of the hot path on one GPU (development aid, not the bench)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt

def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "U"
    M = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1000000
    N = int(float(sys.argv[3])) if len(sys.argv) > 3 else 100000
    res = float(sys.argv[4]) if len(sys.argv) > 4 else 1.0
    t0 = time.time()
    tgt = clouds.target_uniform(M) if kind == "U" else clouds.target_surfaces(M)
    src = clouds.source_from_target(tgt, N)
    print("gen %.2fs" % (time.time() - t0), flush=True)
    g = ndt.NormalDistributionsTransform()
    g.setResolution(res)
    g.setMaximumIterations(28)
    g.setTransformationEpsilon(0.0)
    t0 = time.time(); g.setInputTarget(tgt); t1 = time.time()
    print("setInputTarget first %.1f ms" % ((t1 - t0) * 1e3))
    t0 = time.time(); g.setInputTarget(tgt); t1 = time.time()
    print("setInputTarget again %.1f ms" % ((t1 - t0) * 1e3))
    gi = g.grid()
    print("leaves", len(gi["idx"]), "valid", gi["n_valid"], "div_b", gi["div_b"])
    g.setInputSource(src)
    p = np.zeros(6)
    g.eval(p, True)
    for want_h in (True, False):
        t0 = time.time()
        for _ in range(200):
            r = g.eval(p, want_h)
        dt = (time.time() - t0) / 200
        print("eval(H=%s) %.1f us  nn=%.3f" % (want_h, dt * 1e6, r[3]))
    t0 = time.time()
    for _ in range(50):
        g.hessian_f64(p)
    print("hessian_f64 %.1f us" % ((time.time() - t0) / 50 * 1e6))
    print("server round trip:", g.diag_server_roundtrip(p))
    g.diag_stamps(p)
    st = g.diag_stamps(p).astype(np.int64)
    t0 = st[:, 0].min()
    names = ["entry", "point", "lut", "rec0", "math", "fold", "done"]
    print("stamps (shader cycles, over %d waves): kernel span %d" % (len(st), st[:, 6].max() - t0))
    for k in range(7):
        col = st[:, k] - t0
        d = (st[:, k] - st[:, k - 1]) if k else col
        print("  %-6s abs median %7d max %7d | delta median %6d p90 %6d max %6d" % (names[k], np.median(col), col.max(), np.median(d), np.percentile(d, 90), d.max()))
    g.profile(True); g.profile_read(0); g.profile_read(1)
    g.align()
    n0, ms0 = g.profile_read(0); n1, ms1 = g.profile_read(1)
    g.profile(False)
    print("event-timed K2: with-H %d launches avg %.2f us ; no-H %d launches avg %.2f us" % (n0, ms0 / max(n0, 1) * 1e3, n1, ms1 / max(n1, 1) * 1e3))
    ts = []
    for _ in range(20):
        t0 = time.time(); g.align(); ts.append(time.time() - t0)
    print("align median %.3f ms min %.3f ms" % (np.median(ts) * 1e3, np.min(ts) * 1e3))
    for _ in range(1):
        t0 = time.time(); g.align(); t1 = time.time()
        st = g.stats()
        T = g.getFinalTransformation()
        print("align %.2f ms iters=%d evals=%d hess=%d nn=%.2f  -> %.1f reg/s  rot_err_gt=%.2e trans_err_gt=%.2e" % (
            (t1 - t0) * 1e3, g.getFinalNumIteration(), st["n_evals"], st["n_hessian_recomputes"], st["mean_neighbors"],
            1.0 / (t1 - t0), np.abs(T[:3,:3]-clouds.T_GT_DEFAULT[:3,:3]).max(), np.abs(T[:3,3]-clouds.T_GT_DEFAULT[:3,3]).max()))

if __name__ == "__main__":
    main()
