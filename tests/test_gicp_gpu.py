"""GPU parity tests of the GICP row: the HIP path behind include/gicp_mi355.h against the CPU oracle
(oracle/gicp_oracle.cpp) on the same inputs.  Index work (neighbours, correspondences) must be identical; the f64
covariances agree to rounding; transforms within BASELINE's tolerance (1e-4 rotation / 1e-3 m translation) -- in
practice the whole registration follows the oracle's trajectory evaluation for evaluation."""
import numpy as np
import pytest

from conftest import rot_err, trans_err
from oracle import pyoracle as po
from toyslam_amd import clouds

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-4, 1e-3


@pytest.fixture(scope="module")
def gmod(built_lib):
    assert built_lib.ndt_device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    from toyslam_amd import gicp
    return gicp


@pytest.fixture(scope="module")
def scene():
    tgt = clouds.target_surfaces(20000)[:, :3].astype(np.float32)
    src = clouds.source_from_target(tgt, 8000)[:, :3].astype(np.float32)
    return tgt, src


def both(gmod, tgt, src, **kw):
    g = gmod.GeneralizedIterativeClosestPoint()
    o = po.OracleGICP(**kw)
    if "k" in kw:
        g.setCorrespondenceRandomness(kw["k"])
    if "rotation_epsilon" in kw:
        g.setRotationEpsilon(kw["rotation_epsilon"])
    if "transformation_epsilon" in kw:
        g.setTransformationEpsilon(kw["transformation_epsilon"])
    if "corr_dist_threshold" in kw:
        g.setMaxCorrespondenceDistance(kw["corr_dist_threshold"])
    if "max_iterations" in kw:
        g.setMaximumIterations(kw["max_iterations"])
    if "max_inner_iterations" in kw:
        g.setMaximumOptimizerIterations(kw["max_inner_iterations"])
    for x in (g, o):
        x.setInputTarget(tgt)
        x.setInputSource(src)
    return g, o


def test_neighbours_and_covariances(gmod, scene):
    """computeCovariances (gicp_omp_impl.hpp:48-116): the 20 nearest neighbours of every point are the oracle's,
    in the oracle's order with the oracle's f32 distances; covariances agree to f64 rounding."""
    tgt, src = scene
    g, _ = both(gmod, tgt, src)
    for which, cloud in ((0, tgt), (1, src)):
        cov, idx, d2 = g.covariances(which, neighbors=True)
        oi, od = po.gicp_knn(cloud, cloud, 20)
        assert np.array_equal(idx, oi) and np.array_equal(d2, od)
        assert np.abs(cov - po.gicp_covariances(cloud, 20, 1e-3)).max() < 1e-12
        assert np.array_equal(cov, cov.transpose(0, 2, 1))


@pytest.mark.parametrize("k", [5, 33, 64])
def test_other_neighbourhood_sizes(gmod, scene, k):
    tgt, _ = scene
    sub = tgt[:3000]
    g = gmod.GeneralizedIterativeClosestPoint()
    g.setCorrespondenceRandomness(k)
    g.setInputTarget(sub)
    cov, idx, d2 = g.covariances(0, neighbors=True)
    oi, od = po.gicp_knn(sub, sub, k)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    assert np.abs(cov - po.gicp_covariances(sub, k, 1e-3)).max() < 1e-12


def test_duplicate_points_and_sparse_outliers(gmod):
    """ties are broken by index on both sides; isolated points far from everything (the exhaustive-scan path of the
    shell search) still get their exact neighbours."""
    rng = np.random.default_rng(3)
    base = rng.uniform(-3, 3, (1500, 3)).astype(np.float32)
    cloud = np.concatenate([base, base[:300], base[:100], np.array([[400, 0, 0], [0, -350, 20], [90, 90, 90]], np.float32)])
    g = gmod.GeneralizedIterativeClosestPoint()
    g.setInputTarget(cloud)
    cov, idx, d2 = g.covariances(0, neighbors=True)
    oi, od = po.gicp_knn(cloud, cloud, 20)
    assert np.array_equal(idx, oi) and np.array_equal(d2, od)
    assert np.abs(cov - po.gicp_covariances(cloud, 20, 1e-3)).max() < 1e-9


def test_correspondence_step(gmod, scene):
    """one outer iteration's correspondence step (:405-456): same nearest target index for every source point, same
    gate decisions, Mahalanobis matrices equal as f32 -- with a guess and a non-identity current transform."""
    tgt, src = scene
    guess = clouds.make_T([0.2, -0.1, 0.05], np.radians([0.3, -0.2, 0.6])).astype(np.float32)
    cur = clouds.make_T([0.05, -0.05, 0.02], np.radians([0.1, 0.0, 0.2])).astype(np.float32)
    for thr in (5.0, 0.15):
        g, o = both(gmod, tgt, src, corr_dist_threshold=thr)
        o.prepare(guess)
        m_o, idx_o, maha_o = o.correspond(cur)
        m_g, idx_g, maha_g = g.step_correspond(guess, cur)
        assert m_o == m_g and np.array_equal(idx_o, idx_g)
        if thr < 1:
            assert 0 < m_g < len(src)
        v = idx_o >= 0
        assert np.abs(maha_o[v] - maha_g[v]).max() <= 1e-6 * np.abs(maha_o[v]).max()


def test_functor_sums(gmod, scene):
    """OptimizationFunctorWithIndices (:241-368): operator() (f32 form), df and fdf (f64) at several states."""
    tgt, src = scene
    g, o = both(gmod, tgt, src)
    o.prepare()
    o.correspond(np.eye(4))
    g.step_correspond()
    rng = np.random.default_rng(2)
    for _ in range(6):
        x = np.r_[rng.uniform(-0.3, 0.3, 3), rng.uniform(-0.03, 0.03, 3)]
        for mode in (0, 1, 2):
            fo, go = o.functor(mode, x)
            fg, gg = g.step_functor(mode, x)
            if mode != 1:
                assert abs(fo - fg) <= 1e-12 * abs(fo)
            if mode != 0:
                assert np.abs(go - gg).max() <= 1e-11 * np.abs(go).max()


@pytest.mark.parametrize("case", ["identity", "guess", "k10", "tight_gate", "few_outer", "few_inner"])
def test_align_matches_oracle(gmod, scene, case):
    """computeTransformation (:372-517): final transform within tolerance of the oracle's, same convergence flag,
    same number of outer iterations and of objective evaluations (the optimiser follows the same path)."""
    tgt, src = scene
    kw, guess = {}, None
    if case == "guess":
        guess = clouds.make_T([0.2, -0.1, 0.05], np.radians([0.3, -0.2, 0.6])).astype(np.float32)
    if case == "k10":
        kw = dict(k=10)
    if case == "tight_gate":
        kw = dict(corr_dist_threshold=0.3)
    if case == "few_outer":
        kw = dict(max_iterations=2)
    if case == "few_inner":
        kw = dict(max_inner_iterations=3)
    g, o = both(gmod, tgt, src, **kw)
    ro = o.align(guess, want_cloud=True)
    cloud = g.align(guess, want_cloud=True)
    T = g.getFinalTransformation()
    assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL
    assert g.hasConverged() == ro["converged"] and g.getFinalNumIteration() == ro["iterations"]
    st = g.stats()
    assert (st["n_f"], st["n_df"], st["n_fdf"]) == (ro["n_f"], ro["n_df"], ro["n_fdf"])
    assert st["correspondences"] == ro["correspondences"]
    assert np.abs(cloud - ro["cloud"]).max() < 2e-3
    # the output cloud is the source moved by the final transform ([PCL] transformPointCloud)
    src4 = np.c_[src, np.ones(len(src), np.float32)]
    assert np.array_equal(cloud, po.transform_cloud(src4, T))


def test_caller_supplied_covariances(gmod, scene):
    """setSourceCovariances / setTargetCovariances (gicp_omp.h:165-168,186-189): supplied matrices replace the k-NN
    covariances until the cloud is set again; the registration follows the oracle given the same matrices -- for random
    symmetric positive definite covariances, for the class's own covariances passed back in (same result as computing
    them), and after a reset (cloud set again / None).  A change of k between two aligns does not recompute covariances
    that exist (computeTransformation only computes them while they are empty, gicp_omp_impl.hpp:386-397)."""
    tgt, src = scene
    rng = np.random.default_rng(5)

    def spd(n):
        a = rng.normal(0, 1, (n, 3, 3))
        return (a @ a.transpose(0, 2, 1)) * 0.01 + 1e-3 * np.eye(3)

    def same(g, o, guess=None):
        ro = o.align(guess)
        g.align(guess)
        T = g.getFinalTransformation()
        assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL
        assert g.hasConverged() == ro["converged"] and g.getFinalNumIteration() == ro["iterations"]
        st = g.stats()
        assert (st["n_f"], st["n_df"], st["n_fdf"]) == (ro["n_f"], ro["n_df"], ro["n_fdf"])
        return T

    g, o = both(gmod, tgt, src)
    T_knn = same(g, o)
    cs, ct = spd(len(src)), spd(len(tgt))
    for x in (g, o):
        x.setSourceCovariances(cs)
        x.setTargetCovariances(ct)
    T_user = same(g, o)
    assert trans_err(T_user, T_knn) > 1e-5  # another objective
    assert np.array_equal(g.covariances(1), cs) or np.abs(g.covariances(1) - cs).max() < 1e-15  # returned as supplied
    with pytest.raises(RuntimeError):
        g.setSourceCovariances(cs[:-1])  # one matrix per point
    # the source set again: its covariances are the k-NN ones again, the target keeps the supplied ones
    for x in (g, o):
        x.setInputSource(src)
    same(g, o)
    # the class's own covariances passed back in: the registration of the k-NN covariances
    g2, o2 = both(gmod, tgt, src)
    own_s, own_t = g2.covariances(1), g2.covariances(0)
    g2.setInputSource(src)
    g2.setInputTarget(tgt)
    g2.setSourceCovariances(own_s)
    g2.setTargetCovariances(own_t)
    g2.align()
    assert np.array_equal(g2.getFinalTransformation(), T_knn)
    # None clears; k changed between two aligns: existing covariances stay (reference semantics), new clouds use the new k
    g2.setSourceCovariances(None)
    g2.setTargetCovariances(None)
    g2.align()
    assert np.array_equal(g2.getFinalTransformation(), T_knn)
    g2.setCorrespondenceRandomness(10)
    o2.align()
    o2_k10 = po.OracleGICP(k=10)
    g2.align()
    assert np.array_equal(g2.getFinalTransformation(), T_knn)  # covariances of k = 20 still in place
    g2.setInputSource(src)
    g2.setInputTarget(tgt)
    o2_k10.setInputTarget(tgt)
    o2_k10.setInputSource(src)
    same(g2, o2_k10)


def test_randomised_scenes_follow_the_oracle(gmod):
    """tools/fuzz_gicp.py, short and well-conditioned (structured scenes, default gate): random sizes, k, guesses and
    iteration caps -- identical neighbours and correspondences, the oracle's registration within tolerance."""
    rng = np.random.default_rng(11)
    for case in range(12):
        tgt = clouds.target_surfaces(int(rng.integers(3000, 12000)), seed=int(rng.integers(1 << 30)), extent=float(rng.choice([40.0, 100.0])))[:, :3].astype(np.float32)
        Tm = clouds.random_T(rng, 0.3, 2.0)
        pick = tgt[rng.choice(len(tgt), int(rng.integers(800, 3000)), replace=False)]
        src = (clouds.apply_T(np.linalg.inv(Tm), pick) + rng.normal(0, 0.01, pick.shape)).astype(np.float32)
        kw = dict(k=int(rng.choice([10, 20, 33])), max_iterations=int(rng.choice([3, 200])), max_inner_iterations=int(rng.choice([5, 20])))
        guess = None if case % 2 else clouds.random_T(rng, 0.1, 0.5).astype(np.float32)
        g, o = both(gmod, tgt, src, **kw)
        cov, idx, d2 = g.covariances(1, neighbors=True)
        oi, od = po.gicp_knn(src, src, kw["k"])
        assert np.array_equal(idx, oi) and np.array_equal(d2, od), case
        o.prepare(guess)
        m_o, ci_o, _ = o.correspond(np.eye(4))
        m_g, ci_g, _ = g.step_correspond(guess)
        assert m_o == m_g and np.array_equal(ci_o, ci_g), case
        ro = o.align(guess)
        g.align(guess)
        T = g.getFinalTransformation()
        assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL, (case, kw)
        assert g.hasConverged() == ro["converged"] and g.getFinalNumIteration() == ro["iterations"], (case, kw)


def test_align_on_the_reference_pair(gmod, pair):
    """the bundled scan pair after the 0.1 m prefilter, as ndt_omp/apps/align.cpp:80-86 runs pclomp::GICP on it."""
    tgt, src = pair
    g, o = both(gmod, tgt, src)
    ro = o.align()
    g.align()
    T = g.getFinalTransformation()
    assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL
    assert g.hasConverged() and ro["converged"] and g.getFinalNumIteration() == ro["iterations"]
    # getFitnessScore after align: mean squared nearest-neighbour distance, well under the NDT values of the README
    from scipy.spatial import cKDTree
    moved = clouds.apply_T(T, src)
    want = float(np.mean(cKDTree(tgt.astype(np.float64)).query(moved.astype(np.float64))[0] ** 2))
    assert abs(g.getFitnessScore() - want) < 1e-4 * want
    assert g.getFitnessScore() < 0.25


def test_too_few_correspondences_and_errors(gmod, scene):
    tgt, src = scene
    g, o = both(gmod, tgt, src + np.float32(0.37), corr_dist_threshold=1e-4)
    guess = clouds.make_T([0.5, 0, 0], [0, 0, 0.1]).astype(np.float32)
    ro = o.align(guess)
    g.align(guess)
    assert not g.hasConverged() and not ro["converged"] and g.getFinalNumIteration() == 0
    assert np.array_equal(g.getFinalTransformation(), ro["T"]) and np.array_equal(ro["T"], guess)
    from toyslam_amd import NdtError
    h = gmod.GeneralizedIterativeClosestPoint()
    with pytest.raises(NdtError):  # no inputs
        h.align()
    bad = tgt[:100].copy()
    bad[7, 1] = np.nan
    with pytest.raises(NdtError):  # non-finite input
        h.setInputTarget(bad)
    h.setInputTarget(tgt[:10])
    h.setInputSource(src[:100])
    with pytest.raises(NdtError):  # k_correspondences_ exceeds the target (:53-57)
        h.align()
    with pytest.raises(NdtError):
        h.setCorrespondenceRandomness(65)


def test_handles_are_independent_and_reusable(gmod, scene):
    """new inputs on a used handle drop the cached covariances (gicp_omp.h:128-160); two handles do not interfere."""
    tgt, src = scene
    a, oa = both(gmod, tgt, src)
    b, ob = both(gmod, tgt[::2], src[::3])
    a.align()
    b.align()
    Ta, Tb = a.getFinalTransformation(), b.getFinalTransformation()
    assert trans_err(Ta, oa.align()["T"]) < TRANS_TOL and trans_err(Tb, ob.align()["T"]) < TRANS_TOL
    a.setInputSource(src[::3])
    a.setInputTarget(tgt[::2])
    a.align()
    assert np.array_equal(a.getFinalTransformation(), Tb)


def test_concurrent_handles_from_threads(gmod, scene):
    """four host threads, each with its own handle on the same GPU, registering at the same time: every result equals the
    single-threaded one (handles share nothing but the device pool, which is keyed by stream)."""
    import threading
    tgt, src = scene
    g0 = gmod.GeneralizedIterativeClosestPoint()
    g0.setInputTarget(tgt)
    g0.setInputSource(src)
    g0.align()
    want = g0.getFinalTransformation()
    results, errors = {}, []

    def work(tid):
        try:
            g = gmod.GeneralizedIterativeClosestPoint()
            for rep in range(3):
                g.setInputTarget(tgt)
                g.setInputSource(src)
                g.align()
            results[tid] = g.getFinalTransformation()
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    threads = [threading.Thread(target=work, args=(t,)) for t in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for tid in range(4):
        assert np.array_equal(results[tid], want)


def test_repeated_registrations_are_bit_identical(gmod, pair):
    """the same registration 200 times on one handle, inputs re-set every 20th time: identical transform every time."""
    tgt, src = pair
    g = gmod.GeneralizedIterativeClosestPoint()
    g.setInputTarget(tgt)
    g.setInputSource(src)
    g.align()
    want, st = g.getFinalTransformation(), g.stats()
    for rep in range(200):
        if rep % 20 == 19:
            g.setInputSource(src)
            g.setInputTarget(tgt)
        g.align()
        assert np.array_equal(g.getFinalTransformation(), want) and g.stats() == st, rep


def test_gpu_matches_the_golden_registrations(gmod, pair):
    """the committed golden vectors (tests/golden/gicp_golden.json, from the oracle) without running the oracle: the GPU gives
    the same registrations -- transform within tolerance, same iteration and evaluation counts."""
    import json
    import os
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "gicp_golden.json")))
    t, s = pair
    for name, a in gold["aligns"].items():
        g = gmod.GeneralizedIterativeClosestPoint()
        kw = a["params"]
        if "k" in kw:
            g.setCorrespondenceRandomness(kw["k"])
        if "corr_dist_threshold" in kw:
            g.setMaxCorrespondenceDistance(kw["corr_dist_threshold"])
        if "max_iterations" in kw:
            g.setMaximumIterations(kw["max_iterations"])
        if "max_inner_iterations" in kw:
            g.setMaximumOptimizerIterations(kw["max_inner_iterations"])
        g.setInputTarget(t)
        g.setInputSource(s)
        g.align(None if a["guess"] is None else np.array(a["guess"], np.float32))
        T, want = g.getFinalTransformation(), np.array(a["T"], np.float32)
        assert rot_err(T, want) < ROT_TOL and trans_err(T, want) < TRANS_TOL, name
        st = g.stats()
        assert (g.hasConverged(), g.getFinalNumIteration(), st["n_f"], st["n_df"], st["n_fdf"], st["correspondences"]) == \
               (a["converged"], a["iterations"], a["n_f"], a["n_df"], a["n_fdf"], a["correspondences"]), name


def test_objective_server_survives_a_quiet_host_and_can_be_switched_off(pair):
    """The persistent objective server (one launch per BFGS run): a host that goes quiet for 150 ms in the middle of a run --
    the server's patience is 20 ms, it tells the host and drains on its own -- costs time, not correctness (the launch path
    answers, the next run gets a fresh server); and with NDT_GICP_SERVER=0 every evaluation is its own launch.  Both give the
    registration of the default configuration, bit for bit.  (Subprocesses: the switches are read once per process.)"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, numpy as np; sys.path.insert(0, %r); from toyslam_amd import gicp; "
            "d = np.load(%r); g = gicp.GeneralizedIterativeClosestPoint(); g.setInputTarget(d['target']); g.setInputSource(d['source']); "
            "g.align(); g.align(); print(json.dumps({'T': g.getFinalTransformation().tolist(), 'it': g.getFinalNumIteration(), 'st': g.stats()}))"
            % (root, os.path.join(root, "tests", "golden", "pair_0p1.npz")))
    results = {}
    for name, env in (("default", {}), ("stall", {"NDT_GICP_TEST_STALL_MS": "150"}), ("launches", {"NDT_GICP_SERVER": "0"})):
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=120)
        assert out.returncode == 0, out.stderr[-2000:]
        results[name] = json.loads(out.stdout.strip().splitlines()[-1])
    assert results["stall"] == results["default"] == results["launches"]


def test_align_with_the_aligned_cloud_does_not_wait_for_the_server(gmod, pair):
    """regression: the objective server is told to exit before the aligned cloud is produced and waited for -- otherwise every
    registration sits out the server's 20 ms patience (seen as 80 ms per align in apps/align.cpp)."""
    import time
    t, s = pair
    g = gmod.GeneralizedIterativeClosestPoint()
    g.setInputTarget(t)
    g.setInputSource(s)
    g.align(want_cloud=True)
    times = []
    for _ in range(5):
        t0 = time.perf_counter()
        g.align(want_cloud=True)
        times.append(time.perf_counter() - t0)
    assert np.median(times) < 0.010, times
