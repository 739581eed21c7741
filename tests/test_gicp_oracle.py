"""CPU tests of the GICP row: the oracle restatement of pclomp::GeneralizedIterativeClosestPoint
(oracle/gicp_oracle.cpp) against analytic known answers -- the reference ships neither tests nor published
numbers for GICP, so these are what pins it ("parity unpinned" against the real binary) -- and the
product's host driver (toyslam_amd/csrc/gicp_driver.cpp) against the oracle's, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from oracle import pyoracle as po
from toyslam_amd import clouds

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def scene():
    tgt = clouds.target_surfaces(6000)[:, :3].astype(np.float32)
    src = clouds.source_from_target(tgt, 2500)[:, :3].astype(np.float32)
    return tgt, src


def test_knn_is_exact_and_ordered(scene):
    """[PCL] KdTreeFLANN::nearestKSearch semantics: the k nearest, ascending; checked against scipy's kd-tree."""
    from scipy.spatial import cKDTree
    tgt, src = scene
    idx, d2 = po.gicp_knn(tgt, src[:700], 20)
    dd, ii = cKDTree(tgt.astype(np.float64)).query(src[:700].astype(np.float64), 20)
    assert np.all(np.diff(d2, axis=1) >= 0)
    assert np.abs(d2 - dd ** 2).max() < 1e-5
    assert np.mean([set(a) == set(b) for a, b in zip(idx, ii)]) > 0.995  # f32 vs f64 distance ties at the k-th place
    # ties (duplicated points): ascending index
    dup = np.repeat(tgt[:50], 3, axis=0)
    idx, d2 = po.gicp_knn(dup, dup[:9], 3)
    assert np.array_equal(idx[0], [0, 1, 2]) and np.array_equal(idx[4], [3, 4, 5]) and np.all(d2[:, :3] == 0)


def test_covariances_of_a_plane_closed_form():
    """computeCovariances (gicp_omp_impl.hpp:48-116): on a plane the regularised covariance is
    I - (1 - epsilon) n n^T whatever the in-plane spread."""
    rng = np.random.default_rng(0)
    n = np.array([0.3, -0.5, 0.81]); n /= np.linalg.norm(n)
    a = np.cross(n, [1, 0, 0]); a /= np.linalg.norm(a)
    b = np.cross(n, a)
    uv = rng.uniform(-5, 5, (3000, 2))
    pts = (uv[:, :1] * a + uv[:, 1:] * b + 2.0 * n).astype(np.float32)
    cov = po.gicp_covariances(pts, 20, 1e-3)
    want = np.eye(3) - (1 - 1e-3) * np.outer(n, n)
    assert np.abs(cov - want).max() < 2e-4  # f32 points are not exactly coplanar
    w = np.linalg.eigvalsh(cov)
    assert np.allclose(w, [1e-3, 1, 1], atol=1e-9)
    assert po.gicp_covariances(pts[:10], 20, 1e-3) is None  # k > cloud size: PCL_ERROR and return (:53-57)


def test_apply_state_is_rz_ry_rx():
    """applyState (:519-532): R = Rz(x5) Ry(x4) Rx(x3), translation x0..2, f32."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(1)
    for _ in range(50):
        x = np.r_[rng.uniform(-3, 3, 3), rng.uniform(-1.2, 1.2, 3)]
        T = po.gicp_apply_state(x)
        R = Rotation.from_euler("ZYX", [x[5], x[4], x[3]]).as_matrix()
        assert np.abs(T[:3, :3] - R).max() < 5e-7
        assert np.array_equal(T[:3, 3], x[:3].astype(np.float32)) and np.array_equal(T[3], [0, 0, 0, 1])


def test_functor_gradient_matches_finite_differences(scene):
    """df / fdf (:277-368) against central differences of the f64 objective; operator() (f32 form) agrees with fdf."""
    tgt, src = scene
    g = po.OracleGICP()
    g.setInputTarget(tgt)
    g.setInputSource(src)
    assert g.prepare()
    m, idx, maha = g.correspond(np.eye(4))
    assert m == len(src) and np.all(idx >= 0)
    x = np.array([0.05, -0.02, 0.01, 0.003, -0.002, 0.01])
    f0, _ = g.functor(0, x)
    _, g1 = g.functor(1, x)
    f2, g2 = g.functor(2, x)
    assert abs(f0 - f2) < 1e-6 * abs(f2) and np.array_equal(g1, g2)
    fd = np.zeros(6)
    for i in range(6):
        h = 1e-3 if i < 3 else 1e-4  # T(x) is rounded to f32: steps well above that noise
        xp, xm = x.copy(), x.copy()
        xp[i] += h
        xm[i] -= h
        fd[i] = (g.functor(2, xp)[0] - g.functor(2, xm)[0]) / (2 * h)
    assert np.abs(fd - g2).max() < 2e-3 * np.abs(g2).max()


def test_mahalanobis_matrices(scene):
    """(R C1 R^T + C2)^-1 (:436-452) against numpy on the oracle's own covariances."""
    tgt, src = scene
    g = po.OracleGICP()
    g.setInputTarget(tgt)
    g.setInputSource(src)
    T0 = clouds.make_T([0.1, 0.0, -0.05], [0.01, -0.02, 0.03]).astype(np.float32)
    g.prepare(T0)
    m, idx, maha = g.correspond(np.eye(4))
    c1, c2 = po.gicp_covariances(src, 20, 1e-3), po.gicp_covariances(tgt, 20, 1e-3)
    R = T0[:3, :3].astype(np.float64)
    for i in (0, 7, 100, 2000):
        want = np.linalg.inv(R @ c1[i] @ R.T + c2[idx[i]])
        assert np.abs(maha[i].reshape(3, 3) - want).max() < 1e-4 * np.abs(want).max()


def test_align_recovers_a_known_transform(scene):
    """computeTransformation (:372-517): converges to the generating transform within the noise, with and
    without a guess; the output cloud is the source moved by the final transform."""
    tgt, src = scene
    g = po.OracleGICP()
    g.setInputTarget(tgt)
    g.setInputSource(src)
    r = g.align(want_cloud=True)
    assert r["converged"] and 1 <= r["iterations"] < 50
    assert np.abs(r["T"] - clouds.T_GT_DEFAULT).max() < 0.03
    moved = clouds.apply_T(r["T"], src)
    assert np.abs(r["cloud"][:, :3] - moved).max() < 1e-4
    guess = clouds.make_T([0.25, -0.15, 0.05], np.radians([0.4, -0.2, 0.8])).astype(np.float32)
    r2 = g.align(guess)
    # from a near guess the per-iteration change falls under the epsilons early (plane-to-plane costs barely
    # constrain sliding along the ground): closer than the guess, rotation right, not necessarily at the optimum
    assert r2["converged"]
    assert np.abs(r2["T"] - clouds.T_GT_DEFAULT).max() < np.abs(guess - clouds.T_GT_DEFAULT).max()
    assert np.abs(r2["T"][:3, :3] - clouds.T_GT_DEFAULT[:3, :3]).max() < 1e-3


def test_align_without_enough_correspondences(scene):
    """fewer than 4 correspondences: NotEnoughPointsException -> break, converged_ stays false and the final
    transform is the guess (:476-499, 512)."""
    tgt, src = scene
    g = po.OracleGICP(corr_dist_threshold=1e-4)
    g.setInputTarget(tgt)
    g.setInputSource(src + np.float32(0.37))
    guess = clouds.make_T([0.5, 0, 0], [0, 0, 0.1]).astype(np.float32)
    r = g.align(guess)
    assert not r["converged"] and r["iterations"] == 0
    assert np.array_equal(r["T"], guess)


def test_oracle_reproduces_its_golden_vectors(pair):
    """tests/golden/gicp_golden.json (oracle/gen_golden_gicp.py): regression pins of the restatement on the committed pair --
    neighbours, covariances, one correspondence step, the objective, four full registrations."""
    import json
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "gicp_golden.json")))
    t, s = pair
    cov = po.gicp_covariances(t, 20, 1e-3)
    assert np.allclose(cov.sum(axis=0), gold["target_cov_sum"], rtol=0, atol=1e-7) and np.allclose(cov[0], gold["target_cov_first"], atol=1e-12)
    idx, d2 = po.gicp_knn(t, t[:64], 20)
    assert int(idx.astype(np.int64).sum()) == gold["knn_first64_idx_sum"] and float(d2.astype(np.float64).sum()) == gold["knn_first64_d2_sum"]
    o = po.OracleGICP()
    o.setInputTarget(t)
    o.setInputSource(s)
    o.prepare()
    m, ci, maha = o.correspond(np.eye(4))
    st = gold["step"]
    assert m == st["correspondences"] and int(ci.astype(np.int64).sum()) == st["corr_idx_sum"]
    assert float(maha.astype(np.float64).sum()) == pytest.approx(st["maha_sum"], rel=1e-9)
    assert o.functor(0, st["x"])[0] == pytest.approx(st["f_operator"], rel=1e-12)
    f2, g2 = o.functor(2, st["x"])
    assert f2 == pytest.approx(st["f_fdf"], rel=1e-12) and np.allclose(g2, st["g_fdf"], rtol=1e-10)
    for name, a in gold["aligns"].items():
        og = po.OracleGICP(**a["params"])
        og.setInputTarget(t)
        og.setInputSource(s)
        r = og.align(None if a["guess"] is None else np.array(a["guess"], np.float32))
        assert np.array_equal(r["T"], np.array(a["T"], np.float32)), name
        assert (r["converged"], r["iterations"], r["n_f"], r["n_df"], r["n_fdf"], r["correspondences"]) == \
               (a["converged"], a["iterations"], a["n_f"], a["n_df"], a["n_fdf"], a["correspondences"]), name


def test_product_driver_equals_oracle_driver(tmp_path):
    """tests/gicp_driver_check.cpp: the product's outer loop + BFGS (gicp_driver.cpp) fed the oracle's sums gives
    the oracle's registration bit for bit -- transform, iterations and functor-call counts -- over random scenes,
    guesses, k, iteration caps, distance gates and the too-few-correspondences path."""
    exe = str(tmp_path / "gicp_driver_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-fopenmp", "-msse4.2", "-ffp-contract=off",
                           "-I" + os.path.join(ROOT, "toyslam_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "gicp_driver_check.cpp"), os.path.join(ROOT, "toyslam_amd", "csrc", "gicp_driver.cpp"),
                           os.path.join(ROOT, "oracle", "gicp_oracle.cpp"), os.path.join(ROOT, "oracle", "ndt_oracle.cpp"), "-o", exe])
    out = subprocess.run([exe, "60"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "mismatches 0" in out.stdout


def test_product_driver_under_sanitizers(tmp_path):
    """the same harness built with ASan + UBSan (CPU build only): the BFGS / outer-loop state machine and the oracle's
    restatement run clean over random scenes including the exception path."""
    exe = str(tmp_path / "gicp_driver_check_asan")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fopenmp", "-ffp-contract=off", "-fsanitize=address,undefined",
                           "-fno-sanitize-recover=all", "-I" + os.path.join(ROOT, "toyslam_amd", "csrc"), "-I" + os.path.join(ROOT, "oracle"),
                           os.path.join(ROOT, "tests", "gicp_driver_check.cpp"), os.path.join(ROOT, "toyslam_amd", "csrc", "gicp_driver.cpp"),
                           os.path.join(ROOT, "oracle", "gicp_oracle.cpp"), os.path.join(ROOT, "oracle", "ndt_oracle.cpp"), "-o", exe])
    env = dict(os.environ, OMP_NUM_THREADS="2", ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([exe, "8"], capture_output=True, text=True, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    assert "mismatches 0" in out.stdout


def test_gicp_symbols_exported(built_lib):
    """every entry point include/gicp_mi355.h declares is exported by the library (no compute without a GPU)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "gicp_mi355.h")).read()
    names = set(re.findall(r"\b(gicp_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(built_lib, n), n
