"""GPU: the HIP path (through the C-ABI) against the CPU oracle on the same inputs, against the
committed golden vectors, and -- at BASELINE.json's full size -- through size-independent properties.

Tolerances (north_star): final transform within 1e-4 on rotation entries and 1e-3 m on translation.
Per-evaluation sums are compared far tighter: the f32 per-neighbour math is the same formulae with
FMA contraction on the GPU (<= 1e-6 relative on the f64 sums); integer/index work is bit-exact.
"""
import os

import numpy as np
import pytest

from conftest import ROOT, rot_err, trans_err

pytestmark = pytest.mark.gpu

ROT_TOL, TRANS_TOL = 1e-4, 1e-3


@pytest.fixture(scope="module")
def mods(built_lib):
    assert built_lib.ndt_device_count() >= 1, "no GPU visible: the HIP path cannot run (there is no fallback)"
    from oracle import pyoracle as po
    from toyslam_amd import clouds, ndt
    return ndt, po, clouds


def make_pair(mods, t, s, **kw):
    ndt, po, _ = mods
    g = ndt.NormalDistributionsTransform()
    o = po.OracleNDT(num_threads=8)
    setters = dict(resolution=g.setResolution, step_size=g.setStepSize, trans_eps=g.setTransformationEpsilon,
                   max_iter=g.setMaximumIterations, search_method=g.setNeighborhoodSearchMethod,
                   outlier_ratio=g.setOutlierRatio)
    for k, v in kw.items():
        setters[k](v)
    o.set(**kw)
    g.setInputTarget(t)
    o.set_target(t)
    g.setInputSource(s)
    o.set_source(s)
    return g, o


def launch_path_is_default():
    """False under the development switches that give the launch-per-evaluation path another thread
    partition / operation order than the evaluation server (results then agree to rounding only)."""
    e = os.environ
    return e.get("NDT_K2_FUSED", "1") != "0" and e.get("NDT_SPIN_WAIT", "1") != "0"


def same_transform(a, b):
    if launch_path_is_default():
        return np.array_equal(a, b)
    return rot_err(a, b) < 1e-6 and trans_err(a, b) < 1e-5


def close_sums(a, b, rel=2e-6):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= rel * max(np.abs(b).max(), 1e-30)


def test_native_library_is_in_tree(built_lib):
    import os
    from toyslam_amd import _lib
    assert os.path.dirname(_lib.LIB_PATH).endswith("toyslam_amd") and os.path.exists(_lib.LIB_PATH)


def test_wave_fold_reduction_selftest(mods):
    """The VALU-only reduce-scatter used by every kernel epilogue: exact on half-integers."""
    ndt, _, _ = mods
    nb = 5
    got = ndt.NormalDistributionsTransform().selftest_reduce(nb)
    gt = np.arange(nb * 256, dtype=np.int64)
    for k in range(29):
        vals = 0.5 * ((gt * 131 + k * 17 + (gt >> 3) * k) % 1009).astype(np.float64) - 100.0
        assert np.array_equal(got[:, k], vals.reshape(nb, 256).sum(axis=1)), k


# ------------------------------------------------------------------ K1: voxel grid
def test_grid_bit_exact_indices_and_sums(mods, pair, golden_grid):
    t, s = pair
    g, o = make_pair(mods, t, s)
    gg, og = g.grid(), o.grid()
    assert np.array_equal(gg["idx"], og["idx"]) and np.array_equal(gg["n"], og["n"])      # integer work: exact
    assert np.array_equal(gg["min_b"], og["min_b"]) and np.array_equal(gg["div_b"], og["div_b"])
    assert gg["n_valid"] == int((og["n"] >= 6).sum())
    # index-ordered f64 sums: the mean is bit-identical to the reference's sequential accumulation
    assert np.array_equal(gg["mean"], og["mean"])
    scale = np.abs(og["icov"]).max(axis=(1, 2), keepdims=True) + 1e-300
    # ... and so is the covariance (no FMA in the finalize kernel, same operation order) wherever the
    # eigenvalue inflation did not rebuild it from a -- solver-specific -- eigen-decomposition
    ok = og["n"] >= 6
    plain = ok & (og["evals"][:, 0] >= 0.01 * og["evals"][:, 2])
    assert plain.sum() > 50 and np.array_equal(gg["cov"][plain], og["cov"][plain])
    assert np.abs(gg["cov"] - og["cov"]).max() < 1e-12
    assert (np.abs(gg["icov"] - og["icov"]) / scale).max() < 1e-10
    assert np.allclose(gg["evals"], og["evals"], rtol=1e-10, atol=1e-14)
    # and against the committed golden grid
    assert np.array_equal(gg["idx"], golden_grid["idx"]) and np.array_equal(gg["mean"], golden_grid["mean"])


class _DeviceCopies:
    """float32 arrays copied to HBM through the HIP runtime the library itself is linked to (not torch's bundled copy: in a
    process where the library touched the device first, torch's runtime instance finds no GPU)."""

    def __init__(self):
        import ctypes as C
        import re
        from toyslam_amd import _lib
        assert _lib.lib() is not None
        linked = [m.group(1) for m in re.finditer(r"(/\S*libamdhip64\.so[.\d]*)", open("/proc/self/maps").read()) if "/torch/" not in m.group(1)]
        self.C, self.held = C, []
        self.hip = C.CDLL(linked[0] if linked else "libamdhip64.so")
        self.hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        self.hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        self.hip.hipFree.argtypes = [C.c_void_p]

    def put(self, a):
        a = np.ascontiguousarray(a, dtype=np.float32)
        p = self.C.c_void_p()
        assert self.hip.hipMalloc(self.C.byref(p), max(a.nbytes, 16)) == 0
        assert self.hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        self.held.append(p)
        return p.value

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        for p in self.held:
            self.hip.hipFree(p)


@pytest.mark.parametrize("n,floats,dense", [(1, 3, 1), (777, 3, 0), (20000, 4, 1), (20000, 8, 0), (32768, 3, 0), (5000, 5, 1)])
def test_small_host_clouds_route_equals_device_route(mods, n, floats, dense):
    """Host clouds of up to 32 768 points are repacked and bounded on the host (upload_cloud: no repack kernel, no wait);
    the same records handed over in HBM go through k_repack_bbox.  Both must give the same boxes and the same grid, with NaN /
    inf coordinates, any record stride and garbage behind xyz."""
    ndt, _, _ = mods
    rng = np.random.default_rng(1000 + n + floats)
    rec = rng.uniform(-1e3, 1e3, (n, floats)).astype(np.float32)  # (whatever follows xyz in a record must not matter)
    rec[:, :3] = rng.uniform(-25, 25, (n, 3)).astype(np.float32) * np.array([1, 1, 0.2], np.float32)
    if n > 100:
        rec[3, 0] = np.nan
        if not dense:  # (a dense cloud's box takes infinite coordinates as they are: "voxel grid too large" on either route)
            rec[7, 1] = np.inf
            rec[9, 2] = -np.inf
        rec[11, :3] = np.nan
        rec[n - 1, 1] = np.nan  # the last record (12-B records: read without leaving the buffer)
    gh, gd = ndt.NormalDistributionsTransform(), ndt.NormalDistributionsTransform()
    with _DeviceCopies() as dc:
        gh.setInputTarget(rec, is_dense=bool(dense))
        gd.setInputTargetDevice(dc.put(rec), n, floats * 4, is_dense=bool(dense))
        a, b = gh.grid(), gd.grid()
        for k in ("min_b", "max_b", "div_b", "idx", "n", "mean", "cov", "icov"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        # and as a source: the same evaluation
        src = rec[: max(1, n // 2)].copy()
        gh.setInputSource(src)
        gd.setInputSourceDevice(dc.put(src), len(src), floats * 4)
        p = np.array([0.1, -0.05, 0.02, 0.01, -0.02, 0.03])
        (sh, gh_, Hh, nh), (sd, gd_, Hd, nd_) = gh.eval(p), gd.eval(p)
        assert sh == sd and nh == nd_ and np.array_equal(gh_, gd_) and np.array_equal(Hh, Hd)


@pytest.mark.parametrize("n,dense", [(1, 1), (3000, 0), (70000, 1), (300000, 0)])
def test_clouds_by_reference_equal_copied_clouds(mods, n, dense):
    """ndt_set_input_target_device_ref / ndt_set_input_source_device_ref: 16-byte records in HBM used where they lie (what
    pcl::Registration's ConstPtr inputs are, ndt_omp.h:122-127) -- the same boxes, grid, evaluation and registration as the
    copying entry points, whatever sits in the fourth float; anything but aligned 16-byte device records is refused."""
    ndt, _, clouds = mods
    rng = np.random.default_rng(4242 + n)
    rec = np.zeros((n, 4), np.float32)
    rec[:, :3] = (rng.uniform(-40, 40, (n, 3)) * [1, 1, 0.1]).astype(np.float32)
    rec[:, 3] = rng.uniform(-1e6, 1e6, n).astype(np.float32)  # (not 1: nothing may read it)
    if n > 100 and not dense:
        rec[5, 0] = np.nan
        rec[17, 2] = np.inf
    src = np.zeros((max(1, n // 3), 4), np.float32)
    src[:, :3] = clouds.apply_T(clouds.make_T([0.2, -0.1, 0.05], [0.01, -0.005, 0.02]), np.nan_to_num(rec[: len(src), :3], posinf=0.0))
    src[:, 3] = 7.0
    gc, gr = ndt.NormalDistributionsTransform(), ndt.NormalDistributionsTransform()
    with _DeviceCopies() as dc:
        d_t, d_s = dc.put(rec), dc.put(src)
        gc.setInputTargetDevice(d_t, n, 16, is_dense=bool(dense))
        gr.setInputTargetDeviceRef(d_t, n, is_dense=bool(dense))
        a, b = gc.grid(), gr.grid()
        for k in ("min_b", "max_b", "div_b", "idx", "n", "mean", "cov", "icov"):
            assert np.array_equal(a[k], b[k], equal_nan=True), k
        gc.setInputSourceDevice(d_s, len(src), 16)
        gr.setInputSourceDeviceRef(d_s, len(src))
        p = np.array([0.1, -0.05, 0.02, 0.01, -0.02, 0.03])
        (sa, ga, Ha, na), (sb, gb, Hb, nb) = gc.eval(p), gr.eval(p)
        assert sa == sb and na == nb and np.array_equal(ga, gb) and np.array_equal(Ha, Hb)
        if n >= 3000:
            out_c, out_r = gc.align(n_out=len(src)), gr.align(n_out=len(src))
            assert np.array_equal(gc.getFinalTransformation(), gr.getFinalTransformation())
            assert np.array_equal(out_c, out_r) and np.all(out_r[:, 3] == 1.0)  # the aligned cloud's data[3] = 1 (PCL align pre-amble)
            assert gc.getFitnessScore() == gr.getFitnessScore()
        # a clone shares the borrowed cloud
        g2 = gr.copy()
        assert np.array_equal(g2.grid()["mean"], a["mean"])
        # refused: misaligned records
        with pytest.raises(ndt.NdtError):
            gr.setInputTargetDeviceRef(d_t + 4, max(1, n - 1), is_dense=bool(dense))


def test_record_compaction_is_deferred_and_changes_nothing(mods):
    """A target of more than 65 536 points keeps k1_finalize's record numbering until the grid is registered against a second
    time (or a lock-step batch starts): the compaction then moves records and rewrites the look-up table in place -- the
    evaluation sums, registrations, grid dump and the other paths' answers must be the same bits before and after, and a
    grid shared with a clone is left alone."""
    ndt, po, clouds = mods
    tgt = clouds.target_surfaces(150000, extent=50.0, n_boxes=20)
    src = clouds.source_from_target(tgt, 30000)
    p = np.array([0.25, -0.15, 0.08, 0.008, -0.004, 0.015])
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(tgt)
    g.setInputSource(src)
    e0 = g.eval(p, True)          # records as built
    h0 = g.hessian_f64(p)
    g.align()                     # first registration: still as built
    T1 = g.getFinalTransformation()
    e1 = g.eval(p, True)
    g.align()                     # second registration: compacted first
    T2 = g.getFinalTransformation()
    e2 = g.eval(p, True)
    for a, b in ((e0, e1), (e0, e2)):
        assert a[0] == b[0] and a[3] == b[3] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])
    assert np.array_equal(T1, T2) and np.array_equal(h0, g.hessian_f64(p))
    g.setEvaluationPath(0)
    g.align()
    assert np.array_equal(T1, g.getFinalTransformation())
    g.setEvaluationPath(1)
    o = po.OracleNDT(resolution=1.0, num_threads=16)
    o.set_target(tgt)
    a, b = g.grid(), o.grid()     # the leaf arrays quote the records' new numbers
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["n"], b["n"]) and np.array_equal(a["mean"], b["mean"])
    g.align()
    assert np.array_equal(T1, g.getFinalTransformation())
    assert g.calculateScore(src) == pytest.approx(o.calculate_score(src), rel=1e-11)
    # a batch compacts at once; a clone taken before the second registration keeps the grid as built -- same answers
    gb = ndt.NormalDistributionsTransform()
    gb.setInputTarget(tgt)
    res = gb.alignBatch([src, src[:20000]])
    assert np.array_equal(res["T"][0], T1)
    gc = ndt.NormalDistributionsTransform()
    gc.setInputTarget(tgt)
    gc.setInputSource(src)
    gc.align()
    clone = gc.copy()
    gc.align()
    clone.align()
    assert np.array_equal(gc.getFinalTransformation(), T1) and np.array_equal(clone.getFinalTransformation(), T1)


def test_cu_partitions_and_shared_targets(mods, pair):
    """ndt_set_cu_partition / ndt_share_input_target: a handle that prepares the inputs on the side partition of the CUs, a
    handle that registers on the registration partition with the inputs taken over (no copy, no rebuild) -- the pipelined
    form of the nodes' loop (ndt_omp_mapping_node.cpp:151-169) -- gives the bits of the one-handle loop; partitions can be
    changed on a live handle; the evaluation server and the launch path agree on a partitioned handle as everywhere."""
    import threading
    ndt, po, _ = mods
    t, s = pair
    ref = ndt.NormalDistributionsTransform()
    ref.setInputTarget(t)
    ref.setInputSource(s)
    ref.align()
    T_ref, it_ref = ref.getFinalTransformation(), ref.getFinalNumIteration()
    prep, reg = ndt.NormalDistributionsTransform(), ndt.NormalDistributionsTransform()
    prep.setCuPartition(2)
    reg.setCuPartition(1)
    part, n_side = prep.getCuPartition()
    part_r, n_reg = reg.getCuPartition()
    assert (part, part_r) == (2, 1)
    assert (n_side, n_reg) in ((32, 224), (256, 256))  # masked streams, or the whole device where masks are not to be had
    prep.setInputTarget(t)
    prep.setInputSource(s)
    reg.shareInputTarget(prep)
    reg.shareInputSource(prep)
    a, b = reg.grid(), ref.grid()
    for k in ("idx", "n", "mean", "cov", "icov"):
        assert np.array_equal(a[k], b[k]), k
    reg.align()
    assert np.array_equal(reg.getFinalTransformation(), T_ref) and reg.getFinalNumIteration() == it_ref
    reg.setEvaluationPath(0)
    reg.align()
    assert np.array_equal(reg.getFinalTransformation(), T_ref)
    reg.setEvaluationPath(1)
    # the two really run side by side: registrations in this thread, input preparation in another
    stop, errs, n_prep = threading.Event(), [], [0]

    def preparer():
        try:
            while not stop.is_set():
                prep.setInputTarget(t)
                prep.setInputSource(s)
                n_prep[0] += 1
        except Exception as e:
            errs.append(e)
    th = threading.Thread(target=preparer)
    th.start()
    try:
        for _ in range(50):
            reg.align()
            assert np.array_equal(reg.getFinalTransformation(), T_ref)
    finally:
        stop.set()
        th.join()
    assert not errs and n_prep[0] > 0
    # a live handle changes partition (it moves to the stream of the new partition) and keeps working
    reg.setCuPartition(0)
    reg.align()
    assert np.array_equal(reg.getFinalTransformation(), T_ref)
    reg.setCuPartition(2)
    reg.align()
    assert np.array_equal(reg.getFinalTransformation(), T_ref)
    # a shared target brings its resolution along
    other = ndt.NormalDistributionsTransform()
    other.setResolution(3.0)
    other.shareInputTarget(prep)
    assert other.getResolution() == 1.0


@pytest.mark.parametrize("res", [0.5, 2.0])
def test_grid_other_resolutions(mods, pair, res):
    t, s = pair
    g, o = make_pair(mods, t, s, resolution=res)
    gg, og = g.grid(), o.grid()
    assert np.array_equal(gg["idx"], og["idx"]) and np.array_equal(gg["n"], og["n"])
    assert np.array_equal(gg["mean"], og["mean"])
    scale = np.abs(og["icov"]).max(axis=(1, 2), keepdims=True) + 1e-300
    assert (np.abs(gg["icov"] - og["icov"]) / scale).max() < 1e-9


def test_grid_large_leaves_and_nonfinite(mods):
    """Leaves above the in-thread sort limit, rejected (degenerate) voxels and NaN skipping."""
    ndt, po, _ = mods
    rng = np.random.default_rng(11)
    dense = (rng.random((5000, 3)) * [0.9, 0.9, 0.9] + [3.05, 3.05, 0.05]).astype(np.float32)   # one voxel, 5000 pts
    plane = np.c_[rng.random((400, 2)) * 4, np.full(400, 0.5)].astype(np.float32)               # exactly planar
    line = np.c_[np.linspace(6.1, 6.9, 50), np.full(50, 1.5), np.full(50, 0.5)].astype(np.float32)
    same = np.tile(np.array([[8.5, 8.5, 0.5]], np.float32), (20, 1))                            # zero covariance
    pts = np.concatenate([dense, plane, line, same, rng.random((3000, 3)).astype(np.float32) * [10, 10, 1]])
    pts[::97] = np.nan
    g = ndt.NormalDistributionsTransform()
    o = po.OracleNDT()
    g.setInputTarget(pts, is_dense=False)
    o.set_target(pts, is_dense=False)
    gg, og = g.grid(), o.grid()
    assert np.array_equal(gg["idx"], og["idx"]) and np.array_equal(gg["n"], og["n"])
    small = og["n"] <= 64
    assert np.array_equal(gg["mean"][small], og["mean"][small])
    assert np.allclose(gg["mean"], og["mean"], rtol=1e-14, atol=1e-14)
    ok = og["n"] >= 6
    scale = np.abs(og["icov"]).max(axis=(1, 2), keepdims=True) + 1e-300
    assert (np.abs(gg["icov"] - og["icov"])[ok] / scale[ok]).max() < 1e-8


# ------------------------------------------------------------------ K2: derivatives
@pytest.mark.parametrize("key", ["DIRECT7/zero", "DIRECT7/small", "DIRECT7/large", "DIRECT1/zero", "DIRECT1/small",
                                 "DIRECT26/small", "DIRECT26/large", "KDTREE/zero", "KDTREE/small", "KDTREE/large"])
def test_eval_matches_oracle_and_golden(mods, pair, golden, key):
    ndt, po, _ = mods
    t, s = pair
    method = {"DIRECT7": po.DIRECT7, "DIRECT1": po.DIRECT1, "DIRECT26": po.DIRECT26, "KDTREE": po.KDTREE}[key.split("/")[0]]
    g, o = make_pair(mods, t, s, search_method=method)
    e = golden["evals"][key]
    score, grad, H, nn = g.eval(e["p"], True)
    so, go, Ho, nno = o.eval(e["p"], True)
    assert nn == nno == e["mean_neighbors"]                      # neighbour search: exact
    assert score == pytest.approx(so, rel=1e-6) and score == pytest.approx(e["score"], rel=1e-6)
    assert close_sums(grad, go) and close_sums(H, Ho) and close_sums(H, e["H"])
    assert np.array_equal(H, H.T)
    # compute_hessian = false: same score/gradient (a different kernel instantiation, so the f32
    # FMA contraction may differ in the last ulp of individual terms)
    s2, g2, H2, _ = g.eval(e["p"], False)
    assert s2 == pytest.approx(score, rel=1e-7) and close_sums(g2, grad, rel=1e-7) and H2 is None
    # all-f64 Hessian (computeHessian): f64 means and f64 inverse covariances on the device too (the record's side
    # sector), so only the order of the f64 sums differs from the oracle's serial loop
    assert close_sums(g.hessian_f64(e["p"]), o.hessian_f64(e["p"]), rel=1e-11)


def test_eval_is_deterministic(mods, pair):
    t, s = pair
    g, _ = make_pair(mods, t, s)
    p = [0.4, 0.1, -0.02, 0.004, -0.001, -0.01]
    a = g.eval(p, True)
    for _ in range(3):
        b = g.eval(p, True)
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2])


def test_eval_with_guess_matrix(mods, pair):
    """Initial evaluation of align(guess): cloud moved by the guess matrix, angles from eulerAngles."""
    ndt, po, clouds = mods
    t, s = pair
    g, o = make_pair(mods, t, s)
    G = clouds.make_T([0.3, 0.1, -0.05], np.deg2rad([-0.4, 0.3, 0.8])).astype(np.float32)
    p = ndt.host_matrix_to_pose(G)
    tc = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)], G)
    so, go, Ho, _ = o.eval(p, True, tc)
    sg, gg, Hg, _ = g.eval(p, True, T=G)
    assert sg == pytest.approx(so, rel=1e-6) and close_sums(gg, go) and close_sums(Hg, Ho)


def test_calculate_score_with_non_finite_points(mods, pair):
    """NaN / inf / absurdly far points have no neighbourhood in the reference and add nothing."""
    ndt, po, _ = mods
    t, s = pair
    c = s[:2000].copy()
    c[5] = np.nan
    c[7, 1] = np.inf
    c[9] = 1e30
    for m in (po.DIRECT7, po.KDTREE, po.DIRECT26, po.DIRECT1):
        g = ndt.NormalDistributionsTransform()
        g.setNeighborhoodSearchMethod(m)
        g.setInputTarget(t)
        g.setInputSource(s[:10])
        o = po.OracleNDT(search_method=m)
        o.set_target(t)
        o.set_source(s[:10])
        assert g.calculateScore(c) == pytest.approx(o.calculate_score(c), rel=1e-11)


def test_calculate_score(mods, pair, golden):
    ndt, po, _ = mods
    t, s = pair
    g, o = make_pair(mods, t, s)
    moved = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)], po.pose_to_matrix([0.4, 0.1, -0.02, 0.004, -0.001, -0.01]))
    a = g.calculateScore(moved)
    assert a == pytest.approx(o.calculate_score(moved[:, :3]), rel=1e-11)  # all f64 on both sides (the leaf's f64 icov_)
    assert a == pytest.approx(golden["calculate_score_small_DIRECT7"], rel=1e-6)


def test_eval_linearity_in_points(mods, pair):
    """Size-independent property: the sums over a cloud equal the sum over its two halves."""
    t, s = pair
    g, _ = make_pair(mods, t, s)
    p = [0.4, 0.1, -0.02, 0.004, -0.001, -0.01]
    full = g.eval(p, True)
    h = len(s) // 2
    g.setInputSource(s[:h])
    a = g.eval(p, True)
    g.setInputSource(s[h:])
    b = g.eval(p, True)
    assert full[0] == pytest.approx(a[0] + b[0], rel=1e-12)
    assert np.allclose(full[1], a[1] + b[1], rtol=1e-10, atol=1e-9) and np.allclose(full[2], a[2] + b[2], rtol=1e-10, atol=1e-8)


# ------------------------------------------------------------------ full registration
ALIGN_CASES = ["DIRECT7/default", "DIRECT1/default", "DIRECT7/node_params", "DIRECT7/guess", "DIRECT7/guess_neg_roll",
               "DIRECT7/tight", "DIRECT26/default", "KDTREE/default", "KDTREE/node_params"]


@pytest.mark.parametrize("name", ALIGN_CASES)
def test_align_matches_oracle_and_golden(mods, pair, golden, name):
    ndt, po, _ = mods
    t, s = pair
    a = golden["aligns"][name]
    g, o = make_pair(mods, t, s, search_method=a["method"], trans_eps=a["trans_eps"], max_iter=a["max_iter"],
                     step_size=a["step_size"])
    G = None if a["guess"] is None else np.array(a["guess"], dtype=np.float32)
    ro = o.align(G)
    out = g.align(G, n_out=len(s))
    T = g.getFinalTransformation()
    assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL
    assert rot_err(T, a["T"]) < ROT_TOL and trans_err(T, a["T"]) < TRANS_TOL
    assert g.hasConverged() == ro["converged"] == a["converged"]
    st = g.stats()
    if a["trans_eps"] >= 1e-3:
        # realistic stopping rules: the GPU must walk the oracle's exact path (same trials, same
        # f64-Hessian recomputes).  With epsilon ~ 1e-9 the last line-search decisions compare
        # differences below the f32 rounding noise of the sums, so only the transform is pinned.
        assert g.getFinalNumIteration() == ro["iterations"]
        assert st["n_evals"] == ro["n_evals"] and st["n_hessian_recomputes"] == ro["n_hessian_recomputes"]
    assert g.getTransformationProbability() == pytest.approx(ro["trans_probability"], rel=1e-5)
    # align(output): the source moved by the final transformation, w = 1
    expect = po.transform_cloud(np.c_[s, np.ones(len(s), np.float32)], T)
    assert np.array_equal(out, expect)


def test_readme_fitness_on_gpu(mods, pair, golden):
    """The reference's only known answer, now through the HIP path."""
    from scipy.spatial import cKDTree
    ndt, po, _ = mods
    t, s = pair
    for name, method in (("DIRECT7", po.DIRECT7), ("DIRECT1", po.DIRECT1), ("KDTREE", po.KDTREE)):
        g = ndt.NormalDistributionsTransform()
        g.setResolution(1.0)
        g.setNeighborhoodSearchMethod(method)
        g.setInputTarget(t)
        g.setInputSource(s)
        out = g.align(n_out=len(s))
        d, _ = cKDTree(t.astype(np.float64)).query(out[:, :3].astype(np.float64))
        fit = float(np.mean((d.astype(np.float32) ** 2).astype(np.float64)))
        # the window test_readme_fitness_pins_the_transform (tests/test_oracle_golden.py) measured the pin's power for
        assert fit == pytest.approx(golden["readme_fitness"][name], abs=2e-6)
        # ... and computed by the library itself (row N4: getFitnessScore on the GPU)
        assert g.getFitnessScore() == pytest.approx(golden["readme_fitness"][name], abs=2e-6)
        assert g.getFitnessScore() == pytest.approx(fit, rel=1e-6)


def brute_force_fitness(target, moved, max_range=np.inf):
    """[PCL] getFitnessScore with [FLANN] L2_Simple arithmetic: f32 (dx*dx + dy*dy) + dz*dz, exact minimum."""
    f = np.float32
    best = np.full(len(moved), np.inf, dtype=f)
    for a in range(0, len(target), 4096):
        t = target[a:a + 4096].astype(f)
        dx = moved[:, None, 0] - t[None, :, 0]
        dy = moved[:, None, 1] - t[None, :, 1]
        dz = moved[:, None, 2] - t[None, :, 2]
        d2 = (dx * dx + dy * dy).astype(f) + (dz * dz).astype(f)
        best = np.minimum(best, d2.min(axis=1))
    ok = best.astype(np.float64) <= max_range
    return float(best[ok].astype(np.float64).sum() / ok.sum()) if ok.any() else np.finfo(np.float64).max


def test_fitness_score_exact_nearest_neighbour(mods, pair):
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(31)
    sub = s[rng.choice(len(s), 3000, replace=False)]
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    # (a) after a registration; (b) far off (2 m), so that many nearest points are several shells away
    for guess, iters in ((None, 35), (clouds.make_T([2.0, -1.5, 0.8], np.deg2rad([3.0, 2.0, -8.0])).astype(np.float32), 0)):
        g.setInputSource(sub)
        g.setMaximumIterations(iters)
        moved = g.align(guess, n_out=len(sub))[:, :3]
        ref = brute_force_fitness(t, moved)
        assert g.getFitnessScore() == pytest.approx(ref, rel=1e-12)
        # max_range is compared with the SQUARED distance (PCL's rule)
        ref_r = brute_force_fitness(t, moved, max_range=0.05)
        assert ref_r < ref
        assert g.getFitnessScore(0.05) == pytest.approx(ref_r, rel=1e-12)
    assert g.getFitnessScore(0.0) == np.finfo(np.float64).max  # nothing within range
    # source far outside the target's bounding box and a sparse target: the exhaustive fallback
    g2 = ndt.NormalDistributionsTransform()
    sparse = (rng.uniform(-40, 40, (400, 3)) * [1, 1, 0.1]).astype(np.float32)
    g2.setInputTarget(sparse)
    far = (rng.uniform(-5, 5, (300, 3)) + [150.0, -90.0, 30.0]).astype(np.float32)
    g2.setInputSource(far)
    g2.setMaximumIterations(0)
    moved = g2.align(n_out=len(far))[:, :3]
    assert g2.getFitnessScore() == pytest.approx(brute_force_fitness(sparse, moved), rel=1e-12)
    inside = rng.uniform(-40, 40, (500, 3)).astype(np.float32)
    g2.setInputSource(inside)
    moved = g2.align(n_out=len(inside))[:, :3]
    assert g2.getFitnessScore() == pytest.approx(brute_force_fitness(sparse, moved), rel=1e-12)


@pytest.mark.parametrize("kind", ["uniform", "surfaces"])
def test_synthetic_align_vs_oracle(mods, kind):
    ndt, po, clouds = mods
    tgt = clouds.target_uniform(200000, half=(20.0, 20.0, 5.0)) if kind == "uniform" else \
        clouds.target_surfaces(200000, extent=60.0, n_boxes=25)
    src = clouds.source_from_target(tgt, 20000)
    g, o = make_pair(mods, tgt, src, trans_eps=1e-5, max_iter=30)
    ro = o.align()
    g.align()
    T = g.getFinalTransformation()
    assert rot_err(T, ro["T"]) < ROT_TOL and trans_err(T, ro["T"]) < TRANS_TOL
    assert g.getFinalNumIteration() == ro["iterations"]
    if kind == "surfaces":
        assert rot_err(T, clouds.T_GT_DEFAULT) < 2e-3 and trans_err(T, clouds.T_GT_DEFAULT) < 2e-2


@pytest.mark.parametrize("method", ["KDTREE", "DIRECT26", "DIRECT7", "DIRECT1"])
def test_persistent_server_equals_launch_per_evaluation(mods, pair, method):
    """The persistent evaluation server (its angle tables are computed on the device from six cos/sin
    values, its f64 Hessian and final transform run inside the same kernel) must return exactly what
    the one-launch-per-evaluation path returns: same partition, same fold order, same tables."""
    ndt, po, clouds = mods
    t, s = pair
    res = {}
    for persistent in (True, False):
        g = ndt.NormalDistributionsTransform()
        g.setNeighborhoodSearchMethod(getattr(po, method))
        g.setTransformationEpsilon(1e-9)  # the line search iterates near the optimum -> f64 Hessian recomputes
        g.setMaximumIterations(12)
        g.setEvaluationPath(persistent)
        g.setInputTarget(t)
        g.setInputSource(s)
        out = g.align(n_out=len(s))
        res[persistent] = (g.getFinalTransformation().copy(), g.getFinalNumIteration(), g.getTransformationProbability(),
                           g.stats(), np.asarray(out).copy())
    a, b = res[True], res[False]
    if not launch_path_is_default():
        assert same_transform(a[0], b[0])
        return
    assert np.array_equal(a[0], b[0])
    assert a[1] == b[1] and a[2] == b[2]
    assert a[3]["n_evals"] == b[3]["n_evals"] and a[3]["n_hessian_recomputes"] == b[3]["n_hessian_recomputes"]
    if method == "DIRECT7":
        assert a[3]["n_hessian_recomputes"] >= 1, "case does not reach the in-server f64 Hessian"
    assert np.array_equal(a[4], b[4])


def test_profiled_align_is_the_same_align(mods, pair):
    """bench.py's roofline leg times the kernels with HIP events (ndt_profile_enable): that mode must
    run the same registration, evaluation for evaluation."""
    ndt, po, clouds = mods
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    g.setInputSource(s)
    g.align()
    T0, it0, st0 = g.getFinalTransformation().copy(), g.getFinalNumIteration(), g.stats()
    g.profile(True)
    g.profile_read(0), g.profile_read(1), g.profile_read(2)
    g.align()
    n0, ms0 = g.profile_read(0)
    n1, _ = g.profile_read(1)
    n2, _ = g.profile_read(2)
    g.profile(False)
    assert np.array_equal(g.getFinalTransformation(), T0) and g.getFinalNumIteration() == it0
    assert n0 + n1 == st0["n_evals"] and n2 == st0["n_hessian_recomputes"] and n0 >= 1 and ms0 > 0
    # mode 2: the persistent kernel of the registration between one event pair
    g.setEvaluationPath(True)  # (whatever NDT_PERSISTENT says)
    g.profile(2)
    g.profile_read(3)
    g.align()
    g.align()
    n3, ms3 = g.profile_read(3)
    g.profile(0)
    assert same_transform(g.getFinalTransformation(), T0) and n3 == 2 and 0 < ms3 < 1e3


def test_point_stride_32_and_clone(mods, pair):
    """PointXYZI/XYZRGB are 32-byte records; copies share the device grid (value semantics of the nodes)."""
    ndt, _, _ = mods
    t, s = pair
    t8 = np.zeros((len(t), 8), np.float32)
    t8[:, :3] = t
    t8[:, 4] = 77.0
    s8 = np.zeros((len(s), 8), np.float32)
    s8[:, :3] = s
    a = ndt.NormalDistributionsTransform()
    a.setInputTarget(t8)
    a.setInputSource(s8)
    a.align()
    b = ndt.NormalDistributionsTransform()
    b.setInputTarget(t)
    b.setInputSource(s)
    b.align()
    assert np.array_equal(a.getFinalTransformation(), b.getFinalTransformation())
    c = a.copy()
    assert c.hasConverged() and np.array_equal(c.getFinalTransformation(), a.getFinalTransformation())
    c.setTransformationEpsilon(0.01)
    c.setMaximumIterations(64)
    c.align()
    assert c.getFinalNumIteration() >= a.getFinalNumIteration()
    assert np.array_equal(a.getFinalTransformation(), b.getFinalTransformation())   # the copy did not disturb a


def test_output_records_of_32_bytes(mods, pair):
    """Outputs into caller records wider than 16 bytes (PointXYZI / PointXYZRGB clouds: `out_stride_bytes` 32) -- the aligned
    cloud, the N1 filter's centroids and the N2 map: the first 16 bytes of every record are what the 16-byte form returns,
    the rest of the record is left alone (downloads go through the page-locked staging block and a CPU scatter)."""
    import ctypes as C
    ndt, _, _ = mods
    from toyslam_amd import _lib
    L = _lib.lib()
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    g.setInputSource(s)
    ref = g.align(n_out=len(s))
    # aligned cloud
    out = np.full((len(s), 8), 7.5, np.float32)
    assert L.ndt_align(g._h, None, None, None, None, None, out.ctypes.data, 32) == 0
    assert np.array_equal(out[:, :4], ref) and np.all(out[:, 4:] == 7.5)
    # N1
    ref_f = g.voxelGridFilter(t, 0.5)
    out = np.full((len(t), 8), -3.0, np.float32)
    m = C.c_size_t(0)
    t32 = np.ascontiguousarray(t, dtype=np.float32)
    assert L.ndt_voxel_grid_filter(g._h, t32.ctypes.data, len(t32), 12, 1, C.c_float(0.5), out.ctypes.data, 32, C.byref(m)) == 0
    assert m.value == len(ref_f) and np.array_equal(out[:m.value, :3], ref_f) and np.all(out[:, 4:] == -3.0) and np.all(out[m.value:] == -3.0)
    # N2
    g.mapUpdate(t, None, 0.5)
    ref_m = g.mapGet()
    out = np.full((len(ref_m), 8), 11.0, np.float32)
    assert L.ndt_map_get(g._h, out.ctypes.data, 32) == 0
    assert np.array_equal(out[:, :3], ref_m) and np.all(out[:, 4:] == 11.0)


def test_edge_cases(mods, pair):
    ndt, po, _ = mods
    t, s = pair
    from toyslam_amd import NdtError
    g = ndt.NormalDistributionsTransform()
    with pytest.raises(NdtError):
        g.align()                                              # no inputs
    g.setInputTarget(np.zeros((0, 3), np.float32))             # empty target -> empty grid
    g.setInputSource(s[:100])
    g.align()
    assert g.hasConverged() and g.getFinalNumIteration() == 0
    assert np.array_equal(g.getFinalTransformation(), np.eye(4, dtype=np.float32))
    g.setInputTarget(t)
    g.setInputSource(np.zeros((0, 3), np.float32))             # empty source
    g.align()
    assert g.getFinalNumIteration() == 0
    g.setInputSource(s[:5] + 1000.0)                           # far outside the grid: no neighbours at all
    assert g.eval(np.zeros(6))[0] == 0.0
    g.setInputTarget(t[:5])                                    # every voxel below min_points_per_voxel
    g.setInputSource(s[:100])
    assert g.eval(np.zeros(6))[0] == 0.0
    bad = s[:200].copy()
    bad[3] = np.nan                                            # a NaN source point contributes nothing
    g.setInputTarget(t)
    g.setInputSource(bad)
    o = po.OracleNDT()
    o.set_target(t)
    o.set_source(np.delete(bad, 3, axis=0))
    rg, ro = g.eval(np.zeros(6)), o.eval(np.zeros(6))
    assert rg[0] == pytest.approx(ro[0], rel=1e-6)
    assert close_sums(rg[1], ro[1]) and close_sums(rg[2], ro[2])  # gradient and Hessian too: no 0 x NaN leak
    assert np.isfinite(g.hessian_f64(np.zeros(6))).all()
    g.align()
    ra = o.align()
    assert g.getFinalNumIteration() == ra["iterations"] and g.hasConverged() == ra["converged"]
    g.setNeighborhoodSearchMethod(7)                           # unknown value: the reference's `default:` = DIRECT7
    g.setInputSource(s)
    o.set_source(s)
    assert g.eval(np.zeros(6))[0] == pytest.approx(o.eval(np.zeros(6))[0], rel=1e-6)


def test_kdtree_degenerate_voxels(mods):
    """KDTREE next to degenerate voxels.  Because cov_ is seeded with Identity (trap 1) a zero-spread
    voxel is NOT rejected -- its covariance is (n-1)/n^2 * I -- so both searches use it; the
    rejected-voxel branch of trap 7 needs catastrophic cancellation and is not reachable with sane
    coordinates.  The GPU must agree with the oracle on such voxels in both modes."""
    ndt, po, _ = mods
    rng = np.random.default_rng(3)
    good = (rng.random((400, 3)) * [3.0, 3.0, 1.0]).astype(np.float32)
    same = np.tile(np.array([[5.5, 0.5, 0.5]], np.float32), (20, 1))       # zero spread
    tgt = np.concatenate([good, same])
    src = np.concatenate([good[::5], np.array([[5.4, 0.6, 0.4], [5.6, 0.4, 0.6]], np.float32)])
    for method in (po.KDTREE, po.DIRECT7):
        g, o = make_pair(mods, tgt, src, search_method=method)
        og = o.grid()
        k = int(np.nonzero(og["n"] == 20)[0][0])
        assert np.allclose(og["cov"][k], np.eye(3) * 19.0 / 400.0, atol=1e-9)
        p = [0.01, -0.02, 0.0, 0.0, 0.0, 0.003]
        so, go, Ho, nno = o.eval(p, True)
        sg, gg, Hg, nng = g.eval(p, True)
        assert nng == nno
        assert sg == pytest.approx(so, rel=1e-6) and close_sums(gg, go) and close_sums(Hg, Ho)


def test_set_resolution_rebuild_rule(mods, pair):
    """ndt_omp.h:132-142: setResolution rebuilds the grid only when a source is already set."""
    ndt, po, _ = mods
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    g.setResolution(2.0)                 # no source yet: grid stays at 1.0 m
    assert len(g.grid()["idx"]) == 1098
    g.setInputSource(s)
    g.setResolution(0.5)                 # now it rebuilds
    o = po.OracleNDT(resolution=0.5)
    o.set_target(t)
    assert len(g.grid()["idx"]) == len(o.grid()["idx"])


# ------------------------------------------------------------------ batch (map-build)
def test_batch_equals_individual(mods, pair):
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(4)
    scans, guesses = [], []
    for k in range(5):
        T = clouds.random_T(rng, 0.2, 0.5)
        scans.append(clouds.apply_T(np.linalg.inv(T), s[k::5].copy()))
        guesses.append(np.eye(4, dtype=np.float32) if k % 2 == 0 else clouds.make_T([0.05, 0, 0], [0, 0, 0.002]).astype(np.float32))
    g = ndt.NormalDistributionsTransform()
    g.setTransformationEpsilon(0.01)
    g.setMaximumIterations(40)
    g.setInputTarget(t)
    res = g.alignBatch(scans, guesses)
    for k in range(5):
        g.setInputSource(scans[k])
        g.align(guesses[k])
        # batch and single launches are different instantiations of the same kernel body (FMA
        # contraction may differ in the last ulp of a term): same path, transforms equal to f32 noise
        Ts = g.getFinalTransformation()
        assert rot_err(res["T"][k], Ts) < 1e-6 and trans_err(res["T"][k], Ts) < 1e-6
        assert res["iterations"][k] == g.getFinalNumIteration() and res["converged"][k] == g.hasConverged()
        o = po.OracleNDT(trans_eps=0.01, max_iter=40, num_threads=8)
        o.set_target(t)
        o.set_source(scans[k])
        ro = o.align(guesses[k])
        assert rot_err(res["T"][k], ro["T"]) < ROT_TOL and trans_err(res["T"][k], ro["T"]) < TRANS_TOL


def test_batch_stats_and_step_profile(mods, pair):
    """What bench.py's batch roofline is computed from: after ndt_align_batch, ndt_get_stats counts the scan evaluations of all
    members (the sum of what each member needs alone) and their mean neighbour count; with ndt_profile_enable(1) every lock-step's
    derivative kernels are bracketed by one HIP event pair (slot 0) -- and timing them changes no result."""
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(14)
    scans = [clouds.apply_T(np.linalg.inv(clouds.random_T(rng, 0.2, 0.5)), s[k::4].copy()) for k in range(4)]
    g = ndt.NormalDistributionsTransform()
    g.setTransformationEpsilon(0.01)
    g.setMaximumIterations(40)
    g.setInputTarget(t)
    res = g.alignBatch(scans)
    st = g.stats()
    alone_evals = alone_hess = 0
    longest = 0
    for k in range(4):
        g.setInputSource(scans[k])
        g.align()
        assert g.getFinalNumIteration() == res["iterations"][k]
        sk = g.stats()
        alone_evals += sk["n_evals"]
        alone_hess += sk["n_hessian_recomputes"]
        longest = max(longest, sk["n_evals"] + sk["n_hessian_recomputes"])
    g.alignBatch(scans)
    st = g.stats()
    assert st["n_evals"] == alone_evals and st["n_hessian_recomputes"] == alone_hess
    assert 0.5 < st["mean_neighbors"] <= 7.0
    g.profile(1)
    g.profile_read(0)
    res2 = g.alignBatch(scans)
    n_steps, ms = g.profile_read(0)
    g.profile(0)
    assert n_steps == longest and ms > 0  # lock-step: as many steps as the member with the longest request sequence
    assert np.array_equal(res2["T"], res["T"]) and np.array_equal(res2["iterations"], res["iterations"])


# ------------------------------------------------------------------ BASELINE size
def test_full_size_properties(mods):
    """config[1]'s sizes on the surface scene (set S), 30 Newton passes with eps 0: size-independent properties --
    recovery of the known T_gt, run-to-run bit-identity, linearity of the sums in the points, h-bar in range.
    (The headline itself -- set U, eps 1e-9 -- is compared with the oracle and its golden vectors in
    tests/test_gpu_fullsize.py::test_config1_full_size_follows_the_oracle.)"""
    ndt, po, clouds = mods
    tgt = clouds.target_surfaces(1000000)
    src = clouds.source_from_target(tgt, 100000)
    g = ndt.NormalDistributionsTransform()
    g.setMaximumIterations(28)
    g.setTransformationEpsilon(0.0)
    g.setInputTarget(tgt)
    g.setInputSource(src)
    g.align()
    T1 = g.getFinalTransformation()
    assert g.getFinalNumIteration() == 30 and g.stats()["n_evals"] >= 31
    assert rot_err(T1, clouds.T_GT_DEFAULT) < 5e-4 and trans_err(T1, clouds.T_GT_DEFAULT) < 5e-3
    g.align()
    assert np.array_equal(T1, g.getFinalTransformation())
    p = ndt.host_matrix_to_pose(T1)
    full = g.eval(p, True)
    assert 1.0 < full[3] <= 7.0
    g.setInputSource(src[:40000])
    a = g.eval(p, True)
    g.setInputSource(src[40000:])
    b = g.eval(p, True)
    assert full[0] == pytest.approx(a[0] + b[0], rel=1e-12)
    assert np.allclose(full[2], a[2] + b[2], rtol=1e-10, atol=1e-6)


# ------------------------------------------------------------------ N1: scan prefilter (pcl::VoxelGrid)
def test_voxel_grid_filter(mods, pair):
    """Centroid down-sample on the GPU vs the oracle's restatement of pcl::VoxelGrid::applyFilter
    (same f32 accumulation order -> identical), vs an f64 numpy mean, ordering and edge cases."""
    ndt, po, clouds = mods
    from toyslam_amd import NdtError, _lib
    rng = np.random.default_rng(9)
    g = ndt.NormalDistributionsTransform()
    raw = (rng.standard_normal((200000, 3)) * [20, 20, 2]).astype(np.float32)
    for leaf in (0.1, 0.5, 2.0):
        got = g.voxelGridFilter(raw, leaf)
        ref, ov = po.voxel_grid_filter(raw, leaf)
        assert not ov and got.shape == ref.shape
        assert np.array_equal(got, ref)
        assert np.abs(got - clouds.voxel_downsample(raw, leaf)).max() < 2e-5
    # XYZI-shaped input (32-byte records), non-finite points skipped when !is_dense
    xyzi = np.zeros((len(raw), 8), np.float32)
    xyzi[:, :3] = raw
    xyzi[::1000, 0] = np.nan
    got = g.voxelGridFilter(xyzi, 0.5, is_dense=False)
    ref, _ = po.voxel_grid_filter(xyzi, 0.5, is_dense=False)
    assert np.array_equal(got, ref) and np.isfinite(got).all()
    # empty input; single point
    assert g.voxelGridFilter(np.zeros((0, 3), np.float32), 0.5).shape == (0, 3)
    assert np.array_equal(g.voxelGridFilter(np.array([[1.0, 2.0, 3.0]], np.float32), 0.5), [[1.0, 2.0, 3.0]])
    # index-space overflow: PCL warns and passes the input through
    far = np.array([[0, 0, 0], [1e6, 1e6, 1e6]], np.float32)
    with pytest.raises(NdtError) as e:
        g.voxelGridFilter(far, 0.01)
    assert e.value.status == _lib.NDT_ERR_GRID_OVERFLOW
    assert po.voxel_grid_filter(far, 0.01)[1]
    # the filter feeds NDT exactly like apps/align.cpp: same registration as with the oracle's filter
    t, s = pair
    assert np.array_equal(g.voxelGridFilter(t, 0.5), po.voxel_grid_filter(t, 0.5)[0])


# ------------------------------------------------------------------ N2: global map accumulation
def test_map_accumulation_matches_reference_loop(mods, pair):
    """update_global_map (ndt_omp_mapping_node.cpp:195-211): transformPointCloud + '+=' + VoxelGrid,
    three scans in a row, against the oracle's restatements of the same three PCL calls -- bit for bit."""
    ndt, po, clouds = mods
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    rng = np.random.default_rng(21)
    scans = [t, s, (s + rng.normal(0, 0.05, s.shape)).astype(np.float32)]
    poses = [np.eye(4, dtype=np.float32),
             clouds.make_T([0.30, -0.20, 0.10], np.deg2rad([0.5, -0.3, 1.0])).astype(np.float32),
             clouds.make_T([0.70, -0.35, 0.12], np.deg2rad([0.9, -0.2, 2.1])).astype(np.float32)]
    ref = np.zeros((0, 3), np.float32)
    for leaf in (0.5, 0.2):
        g.mapClear()
        ref = np.zeros((0, 3), np.float32)
        for scan, pose in zip(scans, poses):
            n_map, ov = g.mapUpdate(scan, pose, leaf_size=leaf)
            moved = po.transform_cloud(np.c_[scan[:, :3], np.ones(len(scan), np.float32)], pose)[:, :3]
            ref, ov_ref = po.voxel_grid_filter(np.concatenate([ref, moved]), leaf)
            assert not ov and not ov_ref and n_map == len(ref)
            assert np.array_equal(g.mapGet(), ref)
    # identity pose by default; clearing empties the map; an empty scan leaves the (re-filtered) map as it is
    g.mapClear()
    assert g.mapSize() == 0 and g.mapGet().shape == (0, 3)
    g.mapUpdate(t, None, 0.5)
    assert np.array_equal(g.mapGet(), po.voxel_grid_filter(t, 0.5)[0])
    before = g.mapGet()
    g.mapUpdate(np.zeros((0, 3), np.float32), None, 0.5)
    assert np.array_equal(g.mapGet(), po.voxel_grid_filter(before, 0.5)[0])
    # overflow: PCL keeps the unfiltered concatenation
    g.mapClear()
    far = np.array([[0, 0, 0], [1e6, 1e6, 1e6]], np.float32)
    n_map, ov = g.mapUpdate(far, None, 0.01)
    assert ov and n_map == 2 and np.array_equal(g.mapGet(), far)


# ------------------------------------------------------------------ K1 for small clouds: every regime of the one-launch form
@pytest.mark.parametrize("env", [{}, {"NDT_K1_SMALL_LIST": "8"}, {"NDT_K1_LDS_CAP": "512"}, {"NDT_K1_SMALL_FINISH": "0"},
                                 {"NDT_K1_SMALL": "0"}, {"NDT_K1_SMALL": "0", "NDT_K1_INDEX": "1"}, {"NDT_VF_FROM": "0"},
                                 {"NDT_VF_FROM": "0", "NDT_K1_LDS_CAP": "512"}, {"NDT_VF": "chain"}],
                         ids=["default", "lists_overflow_second_scan", "small_passes", "general_finish_only", "chain", "chain_index_form",
                              "voxel_filter_buckets_at_every_size", "voxel_filter_buckets_small_passes", "voxel_filter_general_chain"])
def test_small_cloud_grid_regimes_against_the_oracle(env):
    """tools/fuzz_grid.py (random shapes up to 60 k points: uniform, clusters of thousands of points per voxel, sheets, lines, km
    offsets, NaN, strides, resolutions, grid parameters -- voxel indices / counts / means / covariances bit for bit against the
    oracle, dense == sparse records, N1 and N2 bit-exact) with the one-launch build of small clouds forced through each of its
    paths: wave lists that overflow (the second scan writes straight to the bucketed cloud), LDS passes of 512 points (multi-pass
    and crowded-cell paths of the finish), every bucket through the chain's finish instead of k1_finish_small -- and the chain
    itself, in its point and its 4-byte-index scatter forms, as the cross-check.  The voxel filter (N1, and N2's) likewise: on
    the bucket front end at every size (by default from 128 k points), with small passes, and on the general chain."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_grid.py"), "77", "14"], env=dict(os.environ, FUZZ_GRID_INDEX="1", **env),
                       capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "grid fuzz done, mismatches: 0" in r.stdout, r.stdout[-1500:]


def test_source_ordering_by_radix_passes_is_the_counting_sorts_order():
    """The ordering of big sources (lattice-cell order, a cell's points in the order they came in, non-finite points dropped)
    by stable radix passes of K1's order-preserving scatter and by the counting sort it replaces: the f64 sums of an evaluation
    over the ordered scan and the registration are the same bits (tools/probes/order_check.py in two processes; sources of
    300 to 90 k points, with NaN coordinates, forced through the ordering at every size)."""
    import subprocess
    import sys
    outs = []
    for extra in (dict(NDT_ORDER="chain"), dict(NDT_ORDER_RADIX_FROM="0")):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probes", "order_check.py"), "small"], env=dict(os.environ, NDT_SORT_SOURCE="1", **extra),
                           capture_output=True, text=True, timeout=280)
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append([ln.split("|")[0] for ln in r.stdout.splitlines() if "eval hash" in ln])
    assert len(outs[0]) == 4 and outs[0] == outs[1], outs


# ------------------------------------------------------------------ scans staged to HBM by the sequence's reading threads
def test_staged_sequence_hands_out_the_files_records(mods, pair, tmp_path):
    """ndt_pcd_sequence_stage / _next_device: the records a sequence copied to the device itself are the file's -- a source
    set from the device address registers like the host records of the same file (same bits), file after file, and a second
    pass over the directory (page-locked buffers taken back from the process-wide list) hands out the same."""
    ndt, _, clouds = mods
    t, s = pair
    rng = np.random.default_rng(11)
    scans = [(s + rng.normal(0, 0.01, s.shape)).astype(np.float32) for _ in range(4)]
    for k, c in enumerate(scans):
        clouds.write_pcd_xyz(str(tmp_path / ("cloud_%d.pcd" % (k + 1))), c)
    g, href = ndt.NormalDistributionsTransform(), ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    href.setInputTarget(t)
    for _pass in range(2):
        seq = ndt.PcdSequence(str(tmp_path))
        seq.stage(0)
        assert seq.poll(0) == 4
        for k in range(4):
            d_ptr, h_ptr, n, dense, num = seq.next_device()
            assert n == len(scans[k]) and dense and num == k + 1
            g.setInputSourceDevice(d_ptr, n, 16)
            g.align()
            href.setInputSource(scans[k])
            href.align()
            assert np.array_equal(g.getFinalTransformation(), href.getFinalTransformation()) and g.getFinalNumIteration() == href.getFinalNumIteration()
        assert seq.next_device() is None
        del seq


# ------------------------------------------------------------------ clouds that stay in HBM (ndt_cloud)
def test_resident_clouds_equal_host_buffers(mods, pair):
    """The node loop's steps with the filtered scan staying in HBM as an ndt_cloud -- prefilter, source of one registration,
    target of the next, map update -- against the same steps through host buffers and against the oracle: the points, the
    grid, the registration and the map are the same bits (the same kernels on the same points)."""
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(5)
    raw_t = np.repeat(t, 3, axis=0) + rng.normal(0, 0.02, (3 * len(t), 3)).astype(np.float32)
    raw_s = np.repeat(s, 3, axis=0) + rng.normal(0, 0.02, (3 * len(s), 3)).astype(np.float32)
    g, href = ndt.NormalDistributionsTransform(), ndt.NormalDistributionsTransform()
    ct, ov = g.voxelGridFilterCloud(raw_t.astype(np.float32), 0.5)
    cs, _ = g.voxelGridFilterCloud(raw_s.astype(np.float32), 0.5)
    ft, fs = po.voxel_grid_filter(raw_t.astype(np.float32), 0.5)[0], po.voxel_grid_filter(raw_s.astype(np.float32), 0.5)[0]
    assert not ov and len(ct) == len(ft) and np.array_equal(ct.numpy(), ft) and np.array_equal(cs.numpy(), fs)
    # registration from the resident clouds == from host copies of the same points == the oracle's
    g.setInputTargetCloud(ct)
    g.setInputSourceCloud(cs)
    g.align()
    href.setInputTarget(ft)
    href.setInputSource(fs)
    href.align()
    assert np.array_equal(g.getFinalTransformation(), href.getFinalTransformation())
    assert g.getFinalNumIteration() == href.getFinalNumIteration() and g.grid_counts() == href.grid_counts()
    o = po.OracleNDT(resolution=1.0, num_threads=8)
    o.set_target(ft)
    o.set_source(fs)
    r = o.align()
    assert rot_err(g.getFinalTransformation(), r["T"]) < 1e-4 and trans_err(g.getFinalTransformation(), r["T"]) < 1e-3
    d1 = g.grid()
    d2 = href.grid()
    for k in ("idx", "n", "mean", "cov", "icov"):
        assert np.array_equal(d1[k], d2[k]), k
    # cloud k of the pair (k-1, k) is the target of the pair (k, k+1): promote, no upload
    g.promoteSourceToTarget()
    g.setInputSourceCloud(ct)
    g.align()
    href.setInputTarget(fs)
    href.setInputSource(ft)
    href.align()
    assert np.array_equal(g.getFinalTransformation(), href.getFinalTransformation())
    # promote after a plain host upload of the source works the same way
    h2 = ndt.NormalDistributionsTransform()
    h2.setInputSource(fs)
    h2.promoteSourceToTarget()
    h2.setInputSource(ft)
    h2.align()
    assert np.array_equal(h2.getFinalTransformation(), href.getFinalTransformation())
    # the map update from the resident cloud
    pose = clouds.make_T([0.30, -0.20, 0.10], np.deg2rad([0.5, -0.3, 1.0])).astype(np.float32)
    g.mapClear()
    href.mapClear()
    g.mapUpdateCloud(ct, None, 0.5)
    n1, _ = g.mapUpdateCloud(cs, pose, 0.5)
    href.mapUpdate(ft, None, 0.5)
    n2, _ = href.mapUpdate(fs, pose, 0.5)
    assert n1 == n2 and np.array_equal(g.mapGet(), href.mapGet())
    # a cloud outlives the caller's reference while a handle uses it; another handle (another stream) may use it too
    cu = g.uploadCloud(fs)
    other = ndt.NormalDistributionsTransform()
    other.setInputTargetCloud(ct)
    other.setInputSourceCloud(cu)
    cu.release()
    ct.release()
    other.align()
    href.setInputTarget(ft)
    href.setInputSource(fs)
    href.align()
    assert np.array_equal(other.getFinalTransformation(), href.getFinalTransformation())
    # a cloud outlives the handle that made it (and a handle that read it)
    maker = ndt.NormalDistributionsTransform()
    cm, _ = maker.voxelGridFilterCloud(raw_t.astype(np.float32), 0.5)
    reader = ndt.NormalDistributionsTransform()
    reader.setInputTargetCloud(cm)
    del reader, maker
    import gc
    gc.collect()
    g.setInputTargetCloud(cm)
    g.setInputSource(fs)
    g.align()
    href.setInputTarget(ft)
    href.setInputSource(fs)
    href.align()
    assert np.array_equal(g.getFinalTransformation(), href.getFinalTransformation())
    cm._owner = g   # (the wrapper downloads through a live handle)
    assert np.array_equal(cm.numpy(), ft)
    cm.release()
    # empty input -> an empty cloud
    ce, _ = g.voxelGridFilterCloud(np.zeros((0, 3), np.float32), 0.5)
    assert len(ce) == 0
    # the prefilter in two halves (queued on the filter stream beside a registration on the handle's): the same cloud, and the
    # registration that ran in between is the same registration
    craw = g.uploadCloud(raw_s.astype(np.float32))
    g.voxelGridFilterBegin(craw, 0.5)
    g.setInputTarget(ft)
    g.setInputSource(fs)
    g.align()
    cb, ovb = g.voxelGridFilterEnd()
    assert not ovb and np.array_equal(cb.numpy(), fs) and np.array_equal(g.getFinalTransformation(), href.getFinalTransformation())
    with pytest.raises(ndt.NdtError):
        g.voxelGridFilterEnd()   # nothing begun
    g.setInputTargetCloud(cb)    # ... and its boxes came with it: usable as a target at once
    g.setInputSource(ft)
    g.align()
    href.setInputTarget(fs)
    href.setInputSource(ft)
    href.align()
    assert np.array_equal(g.getFinalTransformation(), href.getFinalTransformation())


# ------------------------------------------------------------------ configs[4] shape: voxel pyramid
def test_multiresolution_pyramid_matches_oracle(mods):
    """Coarse-to-fine NDT (2.0 -> 1.0 -> 0.5 m), each level's result the next level's guess
    (BASELINE configs[4] at test size): level by level the GPU follows the oracle."""
    ndt, po, clouds = mods
    tgt = clouds.target_surfaces(400000, extent=80.0, n_boxes=40)
    T_gt = clouds.make_T([0.9, -0.6, 0.15], np.deg2rad([1.0, -0.8, 4.0]))
    src = clouds.source_from_target(tgt, 40000, T_gt=T_gt)
    g = ndt.NormalDistributionsTransform()
    o = po.OracleNDT(num_threads=8)
    g.setTransformationEpsilon(1e-3)
    g.setMaximumIterations(40)
    o.set(trans_eps=1e-3, max_iter=40)
    g.setInputSource(src)
    o.set_source(src)
    guess_g = guess_o = None
    for res in (2.0, 1.0, 0.5):
        g.setResolution(res)
        o.set(resolution=res)
        g.setInputTarget(tgt)
        o.set_target(tgt)
        g.align(guess_g)
        r = o.align(guess_o)
        T = g.getFinalTransformation()
        assert rot_err(T, r["T"]) < ROT_TOL and trans_err(T, r["T"]) < TRANS_TOL, "level %.1f m" % res
        assert g.getFinalNumIteration() == r["iterations"]
        guess_g, guess_o = T, r["T"]
    assert rot_err(T, T_gt) < 2e-3 and trans_err(T, T_gt) < 3e-2


def test_server_with_several_points_per_thread(mods):
    """Above 131 072 source points the evaluation server's threads walk more than one point each
    (its grid is capped at one block per CU); the launch path covers the same scan with a different
    partition.  Same registration either way, and the oracle's."""
    ndt, po, clouds = mods
    tgt = clouds.target_surfaces(600000, extent=70.0, n_boxes=40)
    src = clouds.source_from_target(tgt, 300000)
    res = {}
    for persistent in (True, False):
        g = ndt.NormalDistributionsTransform()
        g.setTransformationEpsilon(1e-4)
        g.setEvaluationPath(persistent)
        g.setInputTarget(tgt)
        g.setInputSource(src)
        g.align()
        res[persistent] = (g.getFinalTransformation().copy(), g.getFinalNumIteration(), g.stats()["mean_neighbors"])
    a, b = res[True], res[False]
    assert a[1] == b[1] and a[2] == b[2]
    assert rot_err(a[0], b[0]) < 1e-6 and trans_err(a[0], b[0]) < 1e-5  # partitions differ: sums agree to rounding
    o = po.OracleNDT(num_threads=8, trans_eps=1e-4)
    o.set_target(tgt)
    o.set_source(src)
    r = o.align()
    assert a[1] == r["iterations"]
    assert rot_err(a[0], r["T"]) < ROT_TOL and trans_err(a[0], r["T"]) < TRANS_TOL


def test_concurrent_handles_share_one_gpu(mods, pair):
    """Four host threads, four handles, one GPU: the persistent evaluation servers take turns (each
    needs all its blocks resident), every registration still returns its own bit-identical result."""
    import threading
    ndt, po, clouds = mods
    t, s = pair
    tgt = clouds.target_uniform(300000, half=(30.0, 30.0, 5.0))
    src = clouds.source_from_target(tgt, 120000)

    def make(big):
        g = ndt.NormalDistributionsTransform()
        if big:
            g.setInputTarget(tgt)
            g.setInputSource(src)
        else:
            g.setInputTarget(t)
            g.setInputSource(s)
        g.align()
        return g, g.getFinalTransformation().copy(), g.getFinalNumIteration()

    handles = [make(True), make(False), make(True), make(False)]
    bad = []

    def work(g, T0, it0, reps):
        for _ in range(reps):
            g.align()
            if not (np.array_equal(g.getFinalTransformation(), T0) and g.getFinalNumIteration() == it0):
                bad.append(1)

    threads = [threading.Thread(target=work, args=(g, T0, it0, 60 if i % 2 == 0 else 300)) for i, (g, T0, it0) in enumerate(handles)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=120)
    assert not any(th.is_alive() for th in threads), "a registration did not return"
    assert not bad


def test_resident_clouds_across_threads_and_handles(mods, pair):
    """ndt_cloud objects under four host threads: every thread runs the node loop's steps on a handle of its own (prefilter
    into a cloud, source, a target cloud that ALL threads share and that another handle made, registration, map update,
    release by whichever thread is last), while scan sequences are opened and closed beside them (the page-locked buffer
    list) -- every registration and every map are the one-thread run's bits, nothing hangs."""
    import threading
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(21)
    raw_s = (np.repeat(s, 3, axis=0) + rng.normal(0, 0.02, (3 * len(s), 3))).astype(np.float32)
    maker = ndt.NormalDistributionsTransform()
    shared_target, _ = maker.voxelGridFilterCloud((np.repeat(t, 3, axis=0) + rng.normal(0, 0.02, (3 * len(t), 3))).astype(np.float32), 0.5)

    def loop(g, reps, out):
        for _ in range(reps):
            c, _ov = g.voxelGridFilterCloud(raw_s, 0.5)
            g.setInputTargetCloud(shared_target)
            g.setInputSourceCloud(c)
            g.align()
            g.mapClear()
            n_map, _ = g.mapUpdateCloud(c, g.getFinalTransformation(), 0.5)
            out.append((g.getFinalTransformation().copy(), g.getFinalNumIteration(), n_map, len(c)))
            c.release()

    ref = []
    loop(ndt.NormalDistributionsTransform(), 1, ref)
    outs = [[] for _ in range(4)]
    handles = [ndt.NormalDistributionsTransform() for _ in range(4)]
    threads = [threading.Thread(target=loop, args=(handles[i], 40, outs[i])) for i in range(4)]
    stop = threading.Event()

    def churn_sequences(tmp):
        while not stop.is_set():
            q = ndt.PcdSequence(tmp)
            q.poll(0)
            while q.next_raw() is not None:
                pass
            del q

    import tempfile
    with tempfile.TemporaryDirectory() as tmp:
        for k in range(3):
            clouds.write_pcd_xyz(os.path.join(tmp, "cloud_%d.pcd" % (k + 1)), raw_s[: 20000 + 1000 * k])
        churn = threading.Thread(target=churn_sequences, args=(tmp,))
        churn.start()
        for th in threads:
            th.start()
        for th in threads:
            th.join(timeout=180)
        stop.set()
        churn.join(timeout=60)
    assert not any(th.is_alive() for th in threads) and not churn.is_alive(), "a thread did not return"
    for o in outs:
        assert len(o) == 40
        for T, it, n_map, n_c in o:
            assert np.array_equal(T, ref[0][0]) and (it, n_map, n_c) == ref[0][1:]
    del maker  # (the handle that made the shared target goes first; the cloud outlives it)
    g = ndt.NormalDistributionsTransform()
    g.setInputTargetCloud(shared_target)
    g.setInputSource(s)
    g.align()
    shared_target.release()


# ------------------------------------------------------------------ HBM-resident entry points
def test_device_resident_entry_points_equal_host_ones(mods, pair):
    """Every *_device entry point (clouds handed over as device pointers: prefilter, target, source,
    map update, batch) must do exactly what its host-buffer twin does."""
    import ctypes as C
    ndt, po, clouds = mods
    t, s = pair
    # the HIP runtime the library itself is linked to (a process that has imported torch also holds torch's
    # bundled copy, a second runtime instance that does not own the device)
    import re
    from toyslam_amd import _lib
    linked = [m.group(1) for m in re.finditer(r"(/\S*libamdhip64\.so[.\d]*)", open("/proc/self/maps").read()) if "/torch/" not in m.group(1)]
    hip = C.CDLL(linked[0] if linked else "libamdhip64.so")
    assert _lib.lib() is not None
    hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    hip.hipFree.argtypes = [C.c_void_p]
    held = []

    def to_device(a):  # (N, 4) float32 -> device pointer
        a = np.ascontiguousarray(a, dtype=np.float32)
        p = C.c_void_p()
        rc = hip.hipMalloc(C.byref(p), max(a.nbytes, 16))
        if rc != 0:
            dev, cnt = C.c_int(-1), C.c_int(-1)
            r1, r2 = hip.hipGetDevice(C.byref(dev)), hip.hipGetDeviceCount(C.byref(cnt))
            raise AssertionError("hipMalloc -> %d; hipGetDevice -> %d (%d); hipGetDeviceCount -> %d (%d); lib %s" %
                                 (rc, r1, dev.value, r2, cnt.value, [l.split()[-1] for l in open("/proc/self/maps") if "amdhip64" in l][:4]))
        assert hip.hipMemcpy(p, a.ctypes.data, a.nbytes, 1) == 0  # hipMemcpyHostToDevice
        held.append(p)
        return p.value

    def from_device(ptr, n):
        out = np.zeros((n, 4), np.float32)
        assert hip.hipMemcpy(out.ctypes.data, C.c_void_p(ptr), out.nbytes, 2) == 0  # hipMemcpyDeviceToHost
        return out

    try:
        t4 = np.c_[t, np.ones(len(t), np.float32)]
        s4 = np.c_[s, np.ones(len(s), np.float32)]
        dt, ds = to_device(t4), to_device(s4)
        gh = ndt.NormalDistributionsTransform()   # host twin
        gd = ndt.NormalDistributionsTransform()   # device twin
        # N1
        ref = gh.voxelGridFilter(t, 0.5)
        dout = to_device(np.zeros((len(t), 4), np.float32))
        m = gd.voxelGridFilterDevice(dt, len(t), 16, 0.5, dout)
        assert m == len(ref) and np.array_equal(from_device(dout, m)[:, :3], ref)
        # target / source / align
        gh.setInputTarget(t)
        gh.setInputSource(s)
        gd.setInputTargetDevice(dt, len(t), 16)
        gd.setInputSourceDevice(ds, len(s), 16)
        out_h = gh.align(n_out=len(s))
        gd.align()
        assert np.array_equal(gd.getFinalTransformation(), gh.getFinalTransformation())
        ptr, n = gd.output_device()
        assert n == len(s) and np.array_equal(from_device(ptr, n), out_h)
        # the aligned cloud, still on the device, as the next step's input: N2 ...
        T = gd.getFinalTransformation()
        gh.mapUpdate(t, None, 0.5)
        gh.mapUpdate(s, T, 0.5)
        gd.mapUpdateDevice(dt, len(t), 16, None, 0.5)
        gd.mapUpdateDevice(ds, len(s), 16, T, 0.5)
        assert gd.mapSize() == gh.mapSize() and np.array_equal(gd.mapGet(), gh.mapGet())
        # ... and as a new source (its records are float4, stride 16)
        gd.setInputSourceDevice(ptr, n, 16)
        gh.setInputSource(out_h[:, :3])
        gd.align()
        gh.align()
        assert np.array_equal(gd.getFinalTransformation(), gh.getFinalTransformation())
        # batch
        offs = np.array([0, len(s) // 2, len(s)], dtype=np.uintp)
        rh = gh.alignBatch(clouds=[s[:len(s) // 2], s[len(s) // 2:]])
        rd = gd.alignBatch(device_ptr=ds, offsets=offs, stride_bytes=16)
        for a, b in zip(rh["T"], rd["T"]):
            assert np.array_equal(a, b)
    finally:
        for p in held:
            hip.hipFree(p)


@pytest.mark.parametrize("method", ["DIRECT7", "DIRECT26", "KDTREE"])
def test_ragged_batch_equals_individual(mods, pair, method):
    """A lock-step batch of very different scans -- empty, a handful of points, NaN points, far outside
    the target, big -- in every search mode: each member gets the registration it would get alone."""
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(17)
    nanny = s[:3000].copy()
    nanny[::7] = np.nan
    scans = [s[:2000], np.zeros((0, 3), np.float32), s[2000:2007], nanny, (s[:500] + 500.0).astype(np.float32),
             s, clouds.apply_T(np.linalg.inv(clouds.random_T(rng, 0.3, 1.0)), s[1::2].copy())]
    g = ndt.NormalDistributionsTransform()
    g.setNeighborhoodSearchMethod(getattr(po, method))
    g.setTransformationEpsilon(0.01)
    g.setMaximumIterations(30)
    g.setInputTarget(t)
    res = g.alignBatch(scans)
    # cut into independent groups (one of them holds the scan 500 m away, which changes that group's bounding box):
    # bit for bit the one loop -- a member's point order and block count depend on the member alone
    g.setBatchGroups(3)
    res_g = g.alignBatch(scans)
    g.setBatchGroups(0)
    assert np.array_equal(res["T"], res_g["T"], equal_nan=True) and np.array_equal(res["iterations"], res_g["iterations"])
    assert np.array_equal(res["trans_probability"], res_g["trans_probability"], equal_nan=True)
    for k, scan in enumerate(scans):
        g.setInputSource(scan)
        g.align()
        Ts = g.getFinalTransformation()
        # (a 7-point scan is an ill-conditioned problem: last-ulp differences of the sums are amplified)
        assert rot_err(res["T"][k], Ts) < 1e-5 and trans_err(res["T"][k], Ts) < 1e-5, "scan %d" % k
        assert res["iterations"][k] == g.getFinalNumIteration() and bool(res["converged"][k]) == g.hasConverged(), "scan %d" % k


def test_randomised_registrations_follow_the_oracle(mods, pair):
    """60 random configurations (resolution, search method, step size, outlier ratio, stopping rule,
    cloud subsets with NaN / inf points, random guesses; tools/fuzz_align.py is the long version): the GPU
    registration ends where the oracle's does, in the same number of iterations.  The sums agree to
    ~1e-8, so only an ill-posed case whose line search sits on a tie may take another path: at most one
    such case is tolerated."""
    ndt, po, clouds = mods
    t, s = pair
    rng = np.random.default_rng(2)
    methods = [po.KDTREE, po.DIRECT26, po.DIRECT7, po.DIRECT1]
    off_path = 0
    for case in range(60):
        res = float(rng.choice([0.5, 0.8, 1.0, 1.5, 2.0, 3.0]))
        m = int(rng.choice(methods))
        kw = dict(resolution=res, search_method=m, step_size=float(rng.choice([0.05, 0.1, 0.3])),
                  outlier_ratio=float(rng.choice([0.3, 0.55, 0.8])), trans_eps=float(rng.choice([0.1, 0.01, 1e-3])),
                  max_iter=int(rng.choice([5, 20, 35])))
        nt = int(rng.integers(2000, len(t)))
        ns = int(rng.integers(50, len(s)))
        tt = t[rng.choice(len(t), nt, replace=False)].copy()
        ss = s[rng.choice(len(s), ns, replace=False)].copy()
        dense_t = True
        if rng.random() < 0.3:
            tt[rng.choice(nt, 5, replace=False)] = np.nan
            dense_t = False
        if rng.random() < 0.3:
            ss[rng.choice(ns, 3, replace=False), int(rng.integers(0, 3))] = np.inf if rng.random() < 0.5 else np.nan
        guess = None if rng.random() < 0.5 else clouds.random_T(rng, 0.3, 2.0).astype(np.float32)
        g = ndt.NormalDistributionsTransform()
        o = po.OracleNDT(num_threads=8, **kw)
        g.setResolution(res)
        g.setNeighborhoodSearchMethod(m)
        g.setStepSize(kw["step_size"])
        g.setOutlierRatio(kw["outlier_ratio"])
        g.setTransformationEpsilon(kw["trans_eps"])
        g.setMaximumIterations(kw["max_iter"])
        g.setInputTarget(tt, is_dense=dense_t)
        o.set_target(tt, is_dense=dense_t)
        g.setInputSource(ss)
        o.set_source(ss)
        # the evaluations themselves always agree
        p = np.zeros(6) if guess is None else ndt.host_matrix_to_pose(guess)
        rg, ro = g.eval(p, True), o.eval(p, True)
        assert rg[3] == ro[3] and close_sums(rg[1], ro[1]) and close_sums(rg[2], ro[2]), "case %d" % case
        g.align(guess)
        r = o.align(guess)
        T = g.getFinalTransformation()
        same = (rot_err(T, r["T"]) < ROT_TOL and trans_err(T, r["T"]) < TRANS_TOL and
                g.getFinalNumIteration() == r["iterations"] and g.hasConverged() == r["converged"])
        off_path += 0 if same else 1
    assert off_path <= 1


def test_grid_far_from_the_origin_is_still_the_reference_grid(mods):
    """The reference's covariance formula (_impl.hpp:329-330) cancels catastrophically when the
    coordinates are large against the voxel size (here ~2 km against 2.5 cm voxels): any reordering or
    fused multiply-add in the sums would show in the 7th digit.  The GPU grid follows it bit for bit."""
    ndt, po, _ = mods
    rng = np.random.default_rng(33)
    ctr = rng.uniform(-1.0, 1.0, (6, 3))
    c = (ctr[rng.integers(0, 6, 15000)] + rng.normal(0, 0.003, (15000, 3)) + [1965.0, -1240.0, 310.0]).astype(np.float32)
    g = ndt.NormalDistributionsTransform()
    g.setResolution(0.025)
    g.setMinPointPerVoxel(3)
    g.setInputTarget(c)
    o = po.OracleNDT(resolution=0.025, min_points_per_voxel=3)
    o.set_target(c)
    a, b = g.grid(), o.grid()
    assert np.array_equal(a["idx"], b["idx"]) and np.array_equal(a["n"], b["n"]) and np.array_equal(a["mean"], b["mean"])
    ok = b["n"] >= 3
    plain = ok & (b["evals"][:, 0] >= 0.01 * b["evals"][:, 2])
    assert plain.sum() > 20 and np.array_equal(a["cov"][plain], b["cov"][plain])
    assert b["n"].max() > 64  # voxels beyond the register / insertion sort paths of the finalize kernel
    scale = np.abs(b["icov"][ok]).max()
    assert np.abs(a["icov"][ok] - b["icov"][ok]).max() <= 1e-9 * scale


def test_evaluation_server_gives_up_when_the_host_goes_quiet(mods, pair):
    """Liveness of the persistent kernel: with no command for longer than its patience (20 ms) the server tells the host and
    drains on its own; the next request comes back unserved (no hang), the launch path answers, and a fresh server serves
    again -- all three with the same bits.  A short pause does not disturb it."""
    ndt, po, clouds = mods
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    g.setInputSource(s)
    p = np.array([0.1, -0.05, 0.02, 0.003, -0.002, 0.01])
    # (the launch path in the middle is the one-launch kernel, which cuts the scan into the server's blocks; with
    # NDT_K2_FUSED=0 / NDT_SPIN_WAIT=0 it is the throughput kernel and its own blocks: same sums to the last bits but two)
    same_blocks = os.environ.get("NDT_K2_FUSED", "1") != "0" and os.environ.get("NDT_SPIN_WAIT", "1") != "0"
    served, scores = g.selftest_server_idle(p, 150)
    assert not served and scores[0] == scores[2] and scores[0] != 0
    assert scores[1] == scores[0] if same_blocks else abs(scores[1] - scores[0]) <= 1e-13 * abs(scores[0])
    served, scores = g.selftest_server_idle(p, 2)
    assert served and scores[0] == scores[2]
    assert scores[1] == scores[0] if same_blocks else abs(scores[1] - scores[0]) <= 1e-13 * abs(scores[0])
    g.align()  # and the handle is fine afterwards
    assert g.hasConverged()


# ------------------------------------------------------------------ sparse voxel index (ndt_sparse.hip)
def test_sparse_voxel_index_equals_dense(mods, pair):
    """The sort-built, hash-looked-up voxel index against the dense table on the same clouds: identical voxels, bit-identical
    means / covariances / inverse covariances, bit-identical evaluation sums and registrations -- whatever the search mode."""
    ndt, po, clouds = mods
    t, s = pair
    grids, evals, aligns = {}, {}, {}
    p = np.array([0.3, -0.2, 0.1, 0.01, -0.02, 0.03])
    for mode in (1, 2):
        g = ndt.NormalDistributionsTransform()
        g.setVoxelIndex(mode)
        g.setInputTarget(t)
        g.setInputSource(s)
        grids[mode] = g.grid()
        for method in (ndt.DIRECT7, ndt.DIRECT1, ndt.DIRECT26, ndt.KDTREE):
            g.setNeighborhoodSearchMethod(method)
            evals[mode, method] = g.eval(p, True)
            g.align()
            aligns[mode, method] = (g.getFinalTransformation().copy(), g.getFinalNumIteration())
        assert np.allclose(g.hessian_f64(p), g.hessian_f64(p))
    for k in ("idx", "n", "mean", "cov", "icov", "evals"):
        assert np.array_equal(grids[1][k], grids[2][k]), k
    for method in (ndt.DIRECT7, ndt.DIRECT1, ndt.DIRECT26, ndt.KDTREE):
        a, b = evals[1, method], evals[2, method]
        assert a[0] == b[0] and np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[3] == b[3], method
        assert np.array_equal(aligns[1, method][0], aligns[2, method][0]) and aligns[1, method][1] == aligns[2, method][1]


def test_sparse_index_is_chosen_for_a_mostly_empty_box_and_matches_the_oracle(mods):
    """A map 3 km across at 1 m voxels (4.5e8 cells for 3e5 points): the automatic choice is the sparse index; grid and
    evaluation still equal the oracle's (its std::map does not care about the box either)."""
    ndt, po, clouds = mods
    rng = np.random.default_rng(77)
    centres = rng.uniform([-1500, -1500, 0], [1500, 1500, 40], (300, 3))
    tgt = (centres[rng.integers(0, 300, 300000)] + rng.normal(0, 1.2, (300000, 3))).astype(np.float32)
    src = clouds.source_from_target(tgt, 40000)
    g, o = make_pair(mods, tgt, src)
    gg, og = g.grid(), o.grid()
    assert int(np.prod(gg["div_b"].astype(np.int64))) > 2 ** 25
    assert np.array_equal(gg["idx"], og["idx"]) and np.array_equal(gg["n"], og["n"])
    assert np.array_equal(gg["mean"], og["mean"])
    p = np.array([0.3, -0.2, 0.1, 0.005, -0.003, 0.0175])
    a, b = g.eval(p, True), o.eval(p, True)
    assert a[3] == pytest.approx(b[3], abs=1e-12)
    assert a[0] == pytest.approx(b[0], rel=2e-6) and close_sums(a[1], b[1]) and close_sums(a[2], b[2])
    g.align()
    r = o.align()
    assert rot_err(g.getFinalTransformation(), r["T"]) < ROT_TOL and trans_err(g.getFinalTransformation(), r["T"]) < TRANS_TOL
    assert g.getFinalNumIteration() == r["iterations"]


def test_sparse_prefilter_of_a_wide_scan(mods):
    """pcl::VoxelGrid at 0.1 m over a 300 m wide scan (what apps/align.cpp:60-69 does to every cloud): 5e8 cells for 4e5
    points -- sparse index by the automatic choice -- equal to the dense path's and to the oracle's centroids."""
    ndt, po, clouds = mods
    rng = np.random.default_rng(5)
    scan = clouds.target_surfaces(400000, extent=300.0, n_boxes=80)
    g = ndt.NormalDistributionsTransform()
    ref = po.voxel_grid_filter(scan, 0.1)[0]
    auto = g.voxelGridFilter(scan, 0.1)
    assert np.array_equal(auto, ref)
    g.setVoxelIndex(1)
    dense = g.voxelGridFilter(scan, 0.1)
    assert np.array_equal(dense, ref)
    g.setVoxelIndex(2)
    assert np.array_equal(g.voxelGridFilter(scan[:50000], 0.5), po.voxel_grid_filter(scan[:50000], 0.5)[0])


def test_server_round_diagnostics(mods, pair):
    """The two protocol diagnostics run on the product's buffers and leave the handle usable: the host-driven round trip
    (ndt_diag_server_roundtrip) and the device-driven round (ndt_diag_selfdrive: the last arriving block posts the next
    command itself -- the measurement behind DESIGN.md's "device-side solver" paragraph)."""
    ndt, po, clouds = mods
    t, s = pair
    g = ndt.NormalDistributionsTransform()
    g.setInputTarget(t)
    g.setInputSource(s)
    p = np.array([0.05, -0.03, 0.02, 0.004, -0.002, 0.006])
    before = g.eval(p, True)
    host = g.diag_server_roundtrip(p, 50)
    dev = g.diag_selfdrive(p, 50)
    assert 0.5 < host["nop_us"] < host["with_hessian_us"] < 500.0
    assert 0.5 < dev["protocol_only_us"] < dev["with_hessian_body_us"] < 500.0
    after = g.eval(p, True)
    assert before[0] == after[0] and np.array_equal(before[1], after[1]) and np.array_equal(before[2], after[2])
    g.align()
    assert g.hasConverged()
