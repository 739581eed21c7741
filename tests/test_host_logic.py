"""CPU: the PRODUCT's host-side scalar code (toyslam_amd/csrc/ndt_driver.cpp, via the C-ABI's
ndt_host_* entry points) against the oracle and numpy.  No GPU compute is called."""
import os
import re

import numpy as np
import pytest

from oracle import pyoracle as po

from conftest import ROOT, rot_err, trans_err


@pytest.fixture(scope="module")
def ndt(built_lib):
    from toyslam_amd import ndt as m
    return m


def test_library_exports_every_declared_symbol(built_lib):
    """include/ndt_mi355.h is the boundary: every function it declares must be exported."""
    hdr = open(os.path.join(ROOT, "include", "ndt_mi355.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(ndt_[a-z0-9_]+)\s*\(", hdr)) - {"ndt_allreduce_fn", "ndt_eval_cb"})
    assert len(names) >= 40
    from toyslam_amd import _lib
    for n in names:
        assert hasattr(built_lib, n), "missing export " + n
        assert n in _lib.SIGNATURES, "python binding missing for " + n


def test_no_device_means_error_not_fallback(built_lib, ndt):
    if built_lib.ndt_device_count() > 0:
        pytest.skip("a GPU is present")
    from toyslam_amd import NdtError, _lib
    n = ndt.NormalDistributionsTransform()
    with pytest.raises(NdtError) as e:
        n.setInputTarget(np.zeros((10, 3), np.float32))
    assert e.value.status == _lib.NDT_ERR_NO_DEVICE


def test_host_thread_plan_respects_quota_affinity_and_ranks(ndt):
    """Host threads of a lock-step batch (ndt_batch.hip: StepPool, batch groups) are sized from min(cgroup quota, affinity)
    / LOCAL_WORLD_SIZE: eight ranks on a box that grants 16 CPUs must not start 16 spinning workers each."""
    plan = ndt.host_thread_plan
    assert plan(256, 0.0, 1) == (16, 8)       # a whole host, one rank: the caps
    assert plan(256, 0.0, 8) == (16, 8)       # 32 CPUs per rank
    assert plan(128, 16.0, 1) == (8, 8)       # the one-GPU boxes of this pool: 16 CPUs granted of 128 visible
    assert plan(128, 16.0, 8) == (1, 2)       # the same quota shared by eight ranks: 2 CPUs each
    assert plan(128, 16.7, 8) == (1, 2)       # floor(quota)
    assert plan(8, 0.0, 1) == (4, 8)
    assert plan(8, 0.0, 8) == (1, 1)          # floor 1
    assert plan(4, 2.5, 1) == (1, 2)
    assert plan(1, 0.0, 1) == (1, 1) and plan(0, 0.0, 0) == (1, 1) and plan(64, 0.4, 4) == (1, 1)
    for aff in (1, 3, 16, 100, 512):
        for q in (0.0, 0.5, 1.0, 7.9, 64.0):
            for lw in (1, 2, 8, 64):
                p, g = plan(aff, q, lw)
                cpus = aff if q <= 0 else min(aff, max(1, int(q)))
                share = max(1, cpus // lw)
                assert 1 <= p <= max(1, share // 2) <= 16 or p == 1 or p == 16
                assert 1 <= g <= min(8, share)
    # the probe: what this very process has
    import os
    a, q, w = ndt.host_thread_budget()
    assert a == len(os.sched_getaffinity(0)) and w == int(os.environ.get("LOCAL_WORLD_SIZE", "1")) and q >= 0.0
    try:
        qs, per = open("/sys/fs/cgroup/cpu.max").read().split()
        assert q == (0.0 if qs == "max" else float(qs) / float(per))
    except OSError:
        pass


def test_gauss_constants(ndt):
    for res, ratio in ((1.0, 0.55), (0.5, 0.55), (2.0, 0.35)):
        o = po.OracleNDT(resolution=res, outlier_ratio=ratio)
        assert np.array_equal(ndt.host_gauss(res, ratio), o.gauss())


def test_solve6(ndt):
    rng = np.random.default_rng(0)
    for _ in range(20):
        A = rng.standard_normal((6, 6))
        H = A @ A.T + 0.1 * np.eye(6) + 1e-7 * rng.standard_normal((6, 6))  # slightly asymmetric, like the f32 Hessian
        b = rng.standard_normal(6)
        x = ndt.host_solve6(H, b)
        assert np.allclose(x, np.linalg.solve(H, b), rtol=1e-9, atol=1e-11)
        assert np.allclose(x, po.svd6_solve(H, b), rtol=1e-10, atol=1e-12)
    # rank deficient: minimum-norm least-squares solution, like JacobiSVD::solve
    H = np.diag([3.0, 2.0, 1.0, 0.0, 0.0, 0.0])
    b = np.arange(1.0, 7.0)
    assert np.allclose(ndt.host_solve6(H, b), [1 / 3, 1.0, 3.0, 0, 0, 0])
    Hr = rng.standard_normal((6, 3))
    Hr = Hr @ Hr.T
    assert np.allclose(ndt.host_solve6(Hr, b), np.linalg.pinv(Hr) @ b, atol=1e-9)
    # zero Hessian -> zero step (the reference then returns converged, ndt_omp_impl.hpp:134-139)
    assert np.array_equal(ndt.host_solve6(np.zeros((6, 6)), b), np.zeros(6))
    # NaN propagates so that delta_p_norm != delta_p_norm trips
    Hn = np.eye(6)
    Hn[2, 3] = np.nan
    assert np.isnan(ndt.host_solve6(Hn, b)).any()


def test_chain_pose_is_eigen_f32_product(ndt):
    """pose * transform of the mapping nodes: [Eigen] fixed-size Matrix4f product, no FMA, terms added
    left to right."""
    rng = np.random.default_rng(5)
    f = np.float32
    for _ in range(50):
        a = po.pose_to_matrix(np.r_[rng.uniform(-50, 50, 3), rng.uniform(-3, 3, 3)])
        b = po.pose_to_matrix(np.r_[rng.uniform(-1, 1, 3), rng.uniform(-0.2, 0.2, 3)])
        ref = np.zeros((4, 4), f)
        for i in range(4):
            for j in range(4):
                ref[i, j] = f(f(f(f(a[i, 0] * b[0, j]) + f(a[i, 1] * b[1, j])) + f(a[i, 2] * b[2, j])) + f(a[i, 3] * b[3, j]))
        got = ndt.host_chain_pose(a, b)
        assert np.array_equal(got, ref)
        assert np.allclose(got, a.astype(np.float64) @ b.astype(np.float64), atol=1e-4)
    assert np.array_equal(ndt.host_chain_pose(np.eye(4), a), a)


def test_pose_to_matrix_bit_exact(ndt):
    rng = np.random.default_rng(1)
    for _ in range(200):
        p = np.r_[rng.uniform(-50, 50, 3), rng.uniform(-3.2, 3.2, 3)]
        assert np.array_equal(ndt.host_pose_to_matrix(p), po.pose_to_matrix(p))
    assert np.array_equal(ndt.host_pose_to_matrix(np.zeros(6)), np.eye(4, dtype=np.float32))


def test_matrix_to_pose(ndt):
    rng = np.random.default_rng(2)
    assert np.array_equal(ndt.host_matrix_to_pose(np.eye(4, dtype=np.float32)), np.zeros(6))
    for _ in range(200):
        p = np.r_[rng.uniform(-50, 50, 3), rng.uniform(-1.5, 1.5, 3)]
        T = po.pose_to_matrix(p)
        q = ndt.host_matrix_to_pose(T)
        qo = np.r_[T[:3, 3].astype(np.float64), po.euler_from_matrix(T).astype(np.float64)]
        assert np.array_equal(q, qo)  # same neutral choice for rotation(): the correctly rounded polar factor
        # Eigen's eulerAngles(0,1,2) returns roll in [0, pi]; the pose must map back to the same matrix
        assert -1e-6 <= q[3] <= np.pi + 1e-6
        assert np.abs(ndt.host_pose_to_matrix(q) - T).max() < 5e-6


def test_rotation_substitution_is_bounded_by_an_independent_f32_svd(ndt):
    """rotation() of the guess matrix (Eigen: computeRotationScaling through an f32 JacobiSVD, ndt_omp_impl.hpp:103-111) is
    restated -- by the product and the oracle alike -- as the f64 polar factor rounded to f32, so the two cannot show a
    disagreement with a real f32 SVD.  An f32 SVD of a different family (LAPACK sgesdd through numpy) can: the polar factor
    U V^T it gives, fed through the same Euler extraction, moves the rebuilt guess by less than 2e-6 -- two orders below the
    1e-4 rotation tolerance -- over poses of the size the nodes produce (and the restatement is exact for Identity)."""
    rng = np.random.default_rng(7)
    worst = 0.0
    for _ in range(400):
        p = np.r_[rng.uniform(-50, 50, 3), rng.uniform(-1.5, 1.5, 3)]
        T = po.pose_to_matrix(p)  # f32 products of three f32 axis rotations: orthonormal to ~1e-7
        U, _, Vt = np.linalg.svd(T[:3, :3].astype(np.float32))
        assert U.dtype == np.float32
        T2 = T.copy()
        T2[:3, :3] = (U @ Vt).astype(np.float32)
        q, q2 = ndt.host_matrix_to_pose(T), ndt.host_matrix_to_pose(T2)
        worst = max(worst, float(np.abs(ndt.host_pose_to_matrix(q) - ndt.host_pose_to_matrix(q2)).max()))
    assert worst < 2e-6, worst
    assert np.array_equal(ndt.host_matrix_to_pose(np.eye(4, dtype=np.float32)), np.zeros(6))


def test_angle_derivatives_quirks(ndt):
    p = np.array([0, 0, 0, 0.3, -0.2, 0.5])
    j, h, jd, hd = ndt.host_angle_derivatives(p)
    assert np.allclose(j, jd.astype(np.float32)) and j.dtype == np.float32
    sy = np.sin(p[4])
    assert h[6, 2] == np.float32(sy) and hd[6, 2] == -sy           # trap 3
    hx = h.copy()
    hx[6, 2] = -hx[6, 2]
    assert np.array_equal(hx, hd.astype(np.float32))
    # trap 4: |angle| < 1e-4 snaps to cos = 1, sin = 0
    j0, h0, _, _ = ndt.host_angle_derivatives(np.array([0, 0, 0, 9e-5, -9e-5, 9e-5]))
    jz, hz, _, _ = ndt.host_angle_derivatives(np.zeros(6))
    assert np.array_equal(j0, jz) and np.array_equal(h0, hz)
    cx, sx, cy, cz, sz = np.cos(p[3]), np.sin(p[3]), np.cos(p[4]), np.cos(p[5]), np.sin(p[5])
    assert np.allclose(jd[0], [-sx * sz + cx * sy * cz, -sx * cz - cx * sy * sz, -cx * cy])
    assert np.allclose(hd[14], [-sx * sz + cx * sy * cz, -cx * sy * sz - sx * cz, 0])


CASES = [(None, 0.1, 35), (None, 0.01, 64), (None, 1e-9, 28), ("guess", 0.01, 64), ("guess", 1e-9, 28), (None, 0.0, 6)]


@pytest.mark.parametrize("guess,eps,max_iter", CASES)
def test_product_driver_equals_oracle_driver(ndt, pair, guess, eps, max_iter):
    """The product's Newton/More-Thuente state machine, fed ORACLE evaluations, must walk exactly the
    oracle's path (same trials, same f64 Hessian recomputes, identical final matrix)."""
    from toyslam_amd import clouds
    t, s = pair
    G = None if guess is None else clouds.make_T([0.3, 0.1, -0.05], np.deg2rad([-0.4, 0.3, 0.8])).astype(np.float32)
    o = po.OracleNDT(resolution=1.0, search_method=po.DIRECT7, num_threads=4, trans_eps=eps, max_iter=max_iter)
    o.set_target(t)
    o.set_source(s)
    ref = o.align(G)
    s4 = np.c_[s, np.ones(len(s), np.float32)]

    def evaluator(kind, T, p):
        tc = po.transform_cloud(s4, T)
        if kind == 2:
            o.eval(p, False, tc)  # leaves the angle vectors at p, as the last computeDerivatives did
            return 0.0, np.zeros(6), o.hessian_f64(p)
        sc, g, H, _ = o.eval(p, kind == 0, tc)
        return sc, g, H

    got = ndt.host_run_driver(evaluator, len(s), guess=G, trans_eps=eps, max_iter=max_iter)
    for k in ("converged", "iterations", "n_evals", "n_hessian_recomputes"):
        assert got[k] == ref[k], k
    assert np.array_equal(got["T"], ref["T"])
    assert got["trans_probability"] == ref["trans_probability"]
    if eps == 0.0:
        assert got["iterations"] == max_iter + 2  # trap 9: max passes = max_iterations_ + 2


def test_driver_degenerate_inputs(ndt):
    # zero score / gradient / Hessian (empty target): Newton step is 0 -> converged, identity, 0 iterations
    r = ndt.host_run_driver(lambda k, T, p: (0.0, np.zeros(6), np.zeros((6, 6))), 100)
    assert r["converged"] and r["iterations"] == 0 and np.array_equal(r["T"], np.eye(4, dtype=np.float32))
    # NaN evaluation: not converged (ndt_omp_impl.hpp:134-139)
    r = ndt.host_run_driver(lambda k, T, p: (np.nan, np.full(6, np.nan), np.full((6, 6), np.nan)), 100)
    assert not r["converged"]


def test_quadratic_bowl_converges(ndt):
    """Driver on a synthetic smooth objective: maximise -(p-p*)^T A (p-p*)/2 ; Newton lands on p*."""
    rng = np.random.default_rng(3)
    A = rng.standard_normal((6, 6))
    A = A @ A.T + np.eye(6)
    p_star = np.array([0.05, -0.03, 0.02, 0.01, -0.02, 0.015])

    def evaluator(kind, T, p):
        d = p - p_star
        return -0.5 * d @ A @ d, -(A @ d), -A  # score, gradient, Hessian of the score (negative definite)
    r = ndt.host_run_driver(evaluator, 1, trans_eps=1e-6, max_iter=50, step_size=0.1)
    assert r["converged"]
    assert np.abs(r["T"] - po.pose_to_matrix(p_star)).max() < 1e-5


def test_driver_state_machine_under_sanitizers(tmp_path):
    """tests/driver_fuzz.cpp: the Newton / More-Thuente state machine fed random, singular, zero and
    non-finite evaluation results, built with ASan + UBSan (CPU build only): every run terminates within
    max_iterations + 2 passes and a bounded number of evaluations, without undefined behaviour."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "driver_fuzz")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I" + os.path.join(root, "toyslam_amd", "csrc"),
                           os.path.join(root, "tests", "driver_fuzz.cpp"), os.path.join(root, "toyslam_amd", "csrc", "ndt_driver.cpp"),
                           "-o", exe])
    out = subprocess.run([exe, "5000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert "no crash" in out.stdout


def test_product_driver_equals_oracle_driver_randomised(ndt, pair):
    """tools/fuzz_driver.py, short: random resolutions, search methods, step sizes, outlier ratios, stopping
    rules (incl. epsilon 0 = forced passes), cloud subsets and far guesses -- the product driver fed the
    oracle's evaluations reproduces the oracle's registration bit for bit, evaluation for evaluation."""
    from toyslam_amd import clouds
    t, s = pair
    rng = np.random.default_rng(5)
    for case in range(25):
        kw = dict(resolution=float(rng.choice([0.5, 1.0, 2.0, 3.0])), search_method=int(rng.choice([po.KDTREE, po.DIRECT26, po.DIRECT7, po.DIRECT1])),
                  step_size=float(rng.choice([0.05, 0.1, 0.3])), outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])),
                  trans_eps=float(rng.choice([0.1, 0.01, 1e-3, 1e-9, 0.0])), max_iter=int(rng.choice([0, 3, 12, 35])))
        tt = t[rng.choice(len(t), int(rng.integers(1500, 6000)), replace=False)]
        ss = s[rng.choice(len(s), int(rng.integers(30, 2000)), replace=False)]
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

        got = ndt.host_run_driver(evaluator, len(ss), guess=guess, resolution=kw["resolution"], step_size=kw["step_size"],
                                  outlier_ratio=kw["outlier_ratio"], trans_eps=kw["trans_eps"], max_iter=kw["max_iter"])
        for k in ("converged", "iterations", "n_evals", "n_hessian_recomputes"):
            assert got[k] == ref[k], (case, k)
        assert np.array_equal(got["T"], ref["T"]), case
