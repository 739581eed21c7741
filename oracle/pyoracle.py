"""ctypes front-end of the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  The product package (toyslam_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libndt_oracle.so")

KDTREE, DIRECT26, DIRECT7, DIRECT1 = 0, 1, 2, 3


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in ("ndt_oracle.cpp", "oracle_capi.cpp", "ndt_oracle.hpp", "gicp_oracle.cpp", "gicp_oracle_capi.cpp",
                                           "gicp_oracle.hpp", "Makefile")]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        vp, fp, dp, ip = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.oracle_create.restype = vp
        L.oracle_destroy.argtypes = [vp]
        L.oracle_set_params.argtypes = [vp, C.c_float, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int]
        L.oracle_set_grid_params.argtypes = [vp, C.c_int, C.c_double]
        L.oracle_set_optimised.argtypes = [vp, C.c_int]
        L.oracle_set_target.argtypes = [vp, fp, C.c_size_t, C.c_size_t, C.c_int]
        L.oracle_set_source.argtypes = [vp, fp, C.c_size_t, C.c_size_t]
        L.oracle_align.argtypes = [vp, fp, fp, ip, ip, dp, fp, ip, ip]
        L.oracle_eval.argtypes = [vp, dp, fp, C.c_int, dp, dp, dp, dp]
        L.oracle_hessian_f64.argtypes = [vp, dp, dp]
        L.oracle_calculate_score.argtypes = [vp, fp, C.c_size_t, C.c_size_t]
        L.oracle_calculate_score.restype = C.c_double
        L.oracle_grid_size.argtypes = [vp]
        L.oracle_grid_size.restype = C.c_size_t
        L.oracle_grid_info.argtypes = [vp, ip, ip, ip]
        L.oracle_grid_dump.argtypes = [vp, C.POINTER(C.c_longlong), ip, dp, dp, dp, dp]
        L.oracle_gauss.argtypes = [vp, dp]
        L.oracle_svd6_solve.argtypes = [dp, dp, dp]
        L.oracle_eig3.argtypes = [dp, dp, dp]
        L.oracle_inv3.argtypes = [dp, dp]
        L.oracle_pose_to_matrix.argtypes = [dp, fp]
        L.oracle_euler_from_matrix.argtypes = [fp, fp]
        L.oracle_transform_cloud.argtypes = [fp, C.c_size_t, fp, fp]
        L.oracle_voxel_grid_filter.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_int, C.c_float, fp, ip]
        L.oracle_voxel_grid_filter.restype = C.c_size_t
        L.gicp_oracle_create.restype = vp
        L.gicp_oracle_destroy.argtypes = [vp]
        L.gicp_oracle_set_params.argtypes = [vp, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
        L.gicp_oracle_set_target.argtypes = [vp, fp, C.c_size_t, C.c_size_t]
        L.gicp_oracle_set_source.argtypes = [vp, fp, C.c_size_t, C.c_size_t]
        L.gicp_oracle_covariances.argtypes = [fp, C.c_size_t, C.c_size_t, C.c_int, C.c_double, dp]
        L.gicp_oracle_set_covariances.argtypes = [vp, C.c_int, dp, C.c_size_t]
        L.gicp_oracle_knn.argtypes = [fp, C.c_size_t, fp, C.c_size_t, C.c_int, ip, fp]
        L.gicp_oracle_align.argtypes = [vp, fp, fp, ip, ip, fp]
        L.gicp_oracle_prepare.argtypes = [vp, fp]
        L.gicp_oracle_correspond.argtypes = [vp, fp, ip, fp]
        L.gicp_oracle_functor.argtypes = [vp, C.c_int, dp, dp, dp]
        L.gicp_oracle_apply_state.argtypes = [dp, fp]
        _lib = L
    return _lib


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _xyz(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    assert a.ndim == 2 and a.shape[1] >= 3
    return a


class OracleNDT:
    """Mirrors pclomp::NormalDistributionsTransform's call surface (ndt_omp.h)."""

    def __init__(self, resolution=1.0, step_size=0.1, outlier_ratio=0.55, trans_eps=0.1, max_iter=35,
                 search_method=DIRECT7, num_threads=1, min_points_per_voxel=6, eig_ratio=0.01, optimised=False):
        self.L = lib()
        self.h = C.c_void_p(self.L.oracle_create())
        # optimised=True: the "optimised CPU" timing baseline (dense voxel lookup, per-thread accumulators,
        # parallel f64 Hessian); set before set_target.  Parity tests use the faithful default.
        self.L.oracle_set_optimised(self.h, int(optimised))
        self.params = dict(resolution=resolution, step_size=step_size, outlier_ratio=outlier_ratio,
                           trans_eps=trans_eps, max_iter=max_iter, search_method=search_method,
                           num_threads=num_threads)
        self.L.oracle_set_grid_params(self.h, min_points_per_voxel, eig_ratio)
        self._push()
        self.n_src = 0

    def _push(self):
        p = self.params
        self.L.oracle_set_params(self.h, p["resolution"], p["step_size"], p["outlier_ratio"], p["trans_eps"],
                                 p["max_iter"], p["search_method"], p["num_threads"])

    def set(self, **kw):
        self.params.update(kw)
        self._push()

    def __del__(self):
        try:
            self.L.oracle_destroy(self.h)
        except Exception:
            pass

    def set_target(self, pts, is_dense=True):
        a = _xyz(pts)
        return self.L.oracle_set_target(self.h, _f(a), a.shape[0], a.shape[1], int(is_dense))

    def set_source(self, pts):
        a = _xyz(pts)
        self.n_src = a.shape[0]
        return self.L.oracle_set_source(self.h, _f(a), a.shape[0], a.shape[1])

    def align(self, guess=None, want_cloud=False):
        g = np.eye(4, dtype=np.float32) if guess is None else np.asarray(guess, dtype=np.float32)
        gcm = np.ascontiguousarray(g.T)  # column-major storage
        out = np.zeros(16, dtype=np.float32)
        conv, nit, nev, nh = (C.c_int(0) for _ in range(4))
        tp = C.c_double(0)
        cloud = np.zeros((self.n_src, 4), dtype=np.float32) if want_cloud else None
        self.L.oracle_align(self.h, _f(gcm), _f(out), C.byref(conv), C.byref(nit), C.byref(tp),
                            _f(cloud) if want_cloud else None, C.byref(nev), C.byref(nh))
        res = dict(T=out.reshape(4, 4).T.copy(), converged=bool(conv.value), iterations=nit.value,
                   trans_probability=tp.value, n_evals=nev.value, n_hessian_recomputes=nh.value)
        if want_cloud:
            res["cloud"] = cloud
        return res

    def eval(self, p, compute_hessian=True, trans_cloud=None):
        p = np.ascontiguousarray(p, dtype=np.float64)
        score = C.c_double(0)
        nn = C.c_double(0)
        g = np.zeros(6)
        H = np.zeros(36)
        tc = None
        if trans_cloud is not None:
            tc = np.ascontiguousarray(trans_cloud, dtype=np.float32)
            assert tc.shape == (self.n_src, 4)
        self.L.oracle_eval(self.h, _d(p), _f(tc) if tc is not None else None, int(compute_hessian),
                           C.byref(score), _d(g), _d(H), C.byref(nn))
        return score.value, g, H.reshape(6, 6), nn.value

    def hessian_f64(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        H = np.zeros(36)
        self.L.oracle_hessian_f64(self.h, _d(p), _d(H))
        return H.reshape(6, 6)

    def calculate_score(self, cloud):
        a = _xyz(cloud)
        return self.L.oracle_calculate_score(self.h, _f(a), a.shape[0], a.shape[1])

    def gauss(self):
        d = np.zeros(3)
        self.L.oracle_gauss(self.h, _d(d))
        return d

    def grid(self):
        n = self.L.oracle_grid_size(self.h)
        idx = np.zeros(n, dtype=np.int64)
        npts = np.zeros(n, dtype=np.int32)
        mean = np.zeros((n, 3))
        cov = np.zeros((n, 3, 3))
        icov = np.zeros((n, 3, 3))
        evals = np.zeros((n, 3))
        if n:
            self.L.oracle_grid_dump(self.h, idx.ctypes.data_as(C.POINTER(C.c_longlong)), _i(npts), _d(mean), _d(cov),
                                    _d(icov), _d(evals))
        mb, xb, db = (np.zeros(3, dtype=np.int32) for _ in range(3))
        self.L.oracle_grid_info(self.h, _i(mb), _i(xb), _i(db))
        return dict(idx=idx, n=npts, mean=mean, cov=cov, icov=icov, evals=evals, min_b=mb, max_b=xb, div_b=db)


def svd6_solve(H, b):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(36)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6)
    lib().oracle_svd6_solve(_d(H), _d(b), _d(x))
    return x


def eig3(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(9)
    ev = np.zeros(3)
    V = np.zeros(9)
    lib().oracle_eig3(_d(a), _d(ev), _d(V))
    return ev, V.reshape(3, 3)


def inv3(a):
    a = np.ascontiguousarray(a, dtype=np.float64).reshape(9)
    o = np.zeros(9)
    lib().oracle_inv3(_d(a), _d(o))
    return o.reshape(3, 3)


def pose_to_matrix(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    T = np.zeros(16, dtype=np.float32)
    lib().oracle_pose_to_matrix(_d(p), _f(T))
    return T.reshape(4, 4).T.copy()


def euler_from_matrix(T):
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T)
    a = np.zeros(3, dtype=np.float32)
    lib().oracle_euler_from_matrix(_f(Tc), _f(a))
    return a


def transform_cloud(pts4, T):
    a = np.ascontiguousarray(pts4, dtype=np.float32)
    assert a.shape[1] == 4
    Tc = np.ascontiguousarray(np.asarray(T, dtype=np.float32).T)
    out = np.zeros_like(a)
    lib().oracle_transform_cloud(_f(a), a.shape[0], _f(Tc), _f(out))
    return out


def voxel_grid_filter(pts, leaf, is_dense=True):
    """[PCL] pcl::VoxelGrid centroid down-sample -> ((V,3) float32, overflowed)."""
    a = _xyz(pts)
    out = np.zeros((max(a.shape[0], 1), 4), dtype=np.float32)
    ov = C.c_int(0)
    n = lib().oracle_voxel_grid_filter(_f(a), a.shape[0], a.shape[1], int(is_dense), float(leaf), _f(out), C.byref(ov))
    return out[:n, :3].copy(), bool(ov.value)


class OracleGICP:
    """oracle::GICP (pclomp::GeneralizedIterativeClosestPoint restated on the CPU)."""

    def __init__(self, k=20, gicp_epsilon=1e-3, rotation_epsilon=2e-3, transformation_epsilon=5e-4,
                 corr_dist_threshold=5.0, max_iterations=200, max_inner_iterations=20):
        self._L = lib()
        self._h = self._L.gicp_oracle_create()
        self._L.gicp_oracle_set_params(self._h, k, gicp_epsilon, rotation_epsilon, transformation_epsilon,
                                       corr_dist_threshold, max_iterations, max_inner_iterations)
        self.k, self.eps = k, gicp_epsilon
        self.ns = 0

    def __del__(self):
        if getattr(self, "_h", None):
            self._L.gicp_oracle_destroy(self._h)
            self._h = None

    def setInputTarget(self, pts):
        pts = _xyz(pts)
        self._L.gicp_oracle_set_target(self._h, _f(pts), pts.shape[0], pts.shape[1])

    def setInputSource(self, pts):
        pts = _xyz(pts)
        self.ns = pts.shape[0]
        self._L.gicp_oracle_set_source(self._h, _f(pts), pts.shape[0], pts.shape[1])

    def setTargetCovariances(self, cov):
        """gicp_omp.h:186-189; cov: (n, 3, 3), None clears (they are computed again at the next align)."""
        c = np.zeros((0, 3, 3)) if cov is None else np.ascontiguousarray(cov, dtype=np.float64)
        self._L.gicp_oracle_set_covariances(self._h, 0, _d(c), len(c))

    def setSourceCovariances(self, cov):
        """gicp_omp.h:165-168."""
        c = np.zeros((0, 3, 3)) if cov is None else np.ascontiguousarray(cov, dtype=np.float64)
        self._L.gicp_oracle_set_covariances(self._h, 1, _d(c), len(c))

    def align(self, guess=None, want_cloud=False):
        g = None if guess is None else np.asfortranarray(guess, dtype=np.float32)
        T = np.zeros((4, 4), dtype=np.float32, order="F")
        conv = C.c_int(0)
        stats = np.zeros(5, dtype=np.int32)
        out = np.zeros((self.ns, 4), dtype=np.float32) if want_cloud else None
        self._L.gicp_oracle_align(self._h, None if g is None else _f(g), _f(T), C.byref(conv), _i(stats),
                                  None if out is None else _f(out))
        res = {"T": np.array(T), "converged": bool(conv.value), "iterations": int(stats[0]), "n_f": int(stats[1]),
               "n_df": int(stats[2]), "n_fdf": int(stats[3]), "correspondences": int(stats[4])}
        if want_cloud:
            res["cloud"] = out
        return res

    def prepare(self, guess=None):
        g = np.asfortranarray(np.eye(4) if guess is None else guess, dtype=np.float32)
        return bool(self._L.gicp_oracle_prepare(self._h, _f(g)))

    def correspond(self, transformation):
        t = np.asfortranarray(transformation, dtype=np.float32)
        idx = np.zeros(self.ns, dtype=np.int32)
        maha = np.zeros((self.ns, 9), dtype=np.float32)
        m = self._L.gicp_oracle_correspond(self._h, _f(t), _i(idx), _f(maha))
        return m, idx, maha

    def functor(self, mode, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = C.c_double(0.0)
        g = np.zeros(6)
        self._L.gicp_oracle_functor(self._h, mode, _d(x), C.byref(f), _d(g))
        return f.value, g


def gicp_covariances(pts, k=20, eps=1e-3):
    pts = _xyz(pts)
    out = np.zeros((pts.shape[0], 3, 3))
    ok = lib().gicp_oracle_covariances(_f(pts), pts.shape[0], pts.shape[1], k, eps, _d(out))
    return out if ok else None


def gicp_knn(cloud, query, k):
    cloud = np.ascontiguousarray(np.c_[_xyz(cloud)[:, :3], np.ones(len(cloud), np.float32)], dtype=np.float32)
    query = np.ascontiguousarray(np.c_[_xyz(query)[:, :3], np.ones(len(query), np.float32)], dtype=np.float32)
    idx = np.zeros((len(query), k), dtype=np.int32)
    d2 = np.zeros((len(query), k), dtype=np.float32)
    lib().gicp_oracle_knn(_f(cloud), len(cloud), _f(query), len(query), k, _i(idx), _f(d2))
    return idx, d2


def gicp_apply_state(x):
    T = np.zeros((4, 4), dtype=np.float32, order="F")
    x = np.ascontiguousarray(x, dtype=np.float64)
    lib().gicp_oracle_apply_state(_d(x), _f(T))
    return np.array(T)
