"""Python host-side mirror of pclomp::GeneralizedIterativeClosestPoint over the C-ABI (include/gicp_mi355.h).

Method names follow the reference class (ndt_omp/include/pclomp/gicp_omp.h:52-378) and the
pcl::Registration methods its one caller uses (ndt_omp/apps/align.cpp:14-33,80-86).  All compute runs
in libndt_mi355.so on the GPU; nothing here falls back to numpy.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .ndt import _cloud, _colmajor, _d, _f, _from_colmajor, _i


class GeneralizedIterativeClosestPoint:
    """Drop-in shaped like pclomp::GeneralizedIterativeClosestPoint<PointT, PointT>."""

    def __init__(self, device=0):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.gicp_create(device, C.byref(h)))
        self._h = h
        self._k = 20
        self._n = [0, 0]

    def __del__(self):
        try:
            self._L.gicp_destroy(self._h)
        except Exception:
            pass

    def setCorrespondenceRandomness(self, k):
        check(self._L.gicp_set_correspondence_randomness(self._h, int(k)))
        self._k = int(k)

    def setRotationEpsilon(self, eps):
        check(self._L.gicp_set_rotation_epsilon(self._h, float(eps)))

    def setMaximumOptimizerIterations(self, n):
        check(self._L.gicp_set_maximum_optimizer_iterations(self._h, int(n)))

    def setTransformationEpsilon(self, eps):
        check(self._L.gicp_set_transformation_epsilon(self._h, float(eps)))

    def setMaximumIterations(self, n):
        check(self._L.gicp_set_maximum_iterations(self._h, int(n)))

    def setMaxCorrespondenceDistance(self, d):
        check(self._L.gicp_set_max_correspondence_distance(self._h, float(d)))

    def setInputTarget(self, cloud):
        c = _cloud(cloud)
        check(self._L.gicp_set_input_target(self._h, c.ctypes.data, c.shape[0], c.strides[0]))
        self._n[0] = c.shape[0]

    def setInputSource(self, cloud):
        c = _cloud(cloud)
        check(self._L.gicp_set_input_source(self._h, c.ctypes.data, c.shape[0], c.strides[0]))
        self._n[1] = c.shape[0]

    def setSourceCovariances(self, cov):
        """gicp_omp.h:165-168.  cov: (n, 3, 3) symmetric matrices, one per source point; None clears them."""
        c = None if cov is None else np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9)
        check(self._L.gicp_set_source_covariances(self._h, None if c is None else _d(c), 0 if c is None else len(c)))

    def setTargetCovariances(self, cov):
        """gicp_omp.h:186-189."""
        c = None if cov is None else np.ascontiguousarray(cov, dtype=np.float64).reshape(-1, 9)
        check(self._L.gicp_set_target_covariances(self._h, None if c is None else _d(c), 0 if c is None else len(c)))

    def align(self, guess=None, want_cloud=False):
        g = None if guess is None else _colmajor(guess)
        T = np.zeros(16, dtype=np.float32)
        conv, it = C.c_int(0), C.c_int(0)
        out = np.zeros((self._n[1], 4), dtype=np.float32) if want_cloud else None
        check(self._L.gicp_align(self._h, None if g is None else _f(g), _f(T), C.byref(conv), C.byref(it),
                                 None if out is None else out.ctypes.data))
        return out

    def _result(self):
        T = np.zeros(16, dtype=np.float32)
        conv, it = C.c_int(0), C.c_int(0)
        check(self._L.gicp_get_result(self._h, _f(T), C.byref(conv), C.byref(it)))
        return _from_colmajor(T), bool(conv.value), it.value

    def hasConverged(self):
        return self._result()[1]

    def getFinalTransformation(self):
        return self._result()[0]

    def getFinalNumIteration(self):
        return self._result()[2]

    def getFitnessScore(self, max_range=np.finfo(np.float64).max):
        v = C.c_double(0.0)
        check(self._L.gicp_get_fitness_score(self._h, float(max_range), C.byref(v)))
        return v.value

    def stats(self):
        a = [C.c_int(0) for _ in range(4)]
        check(self._L.gicp_get_stats(self._h, *[C.byref(x) for x in a]))
        return {"n_f": a[0].value, "n_df": a[1].value, "n_fdf": a[2].value, "correspondences": a[3].value}

    # --- inspection (parity tests) ---
    def covariances(self, which, neighbors=False):
        n = self._n[which]
        cov = np.zeros((n, 3, 3))
        if neighbors:
            idx = np.zeros((n, self._k), dtype=np.int32)
            d2 = np.zeros((n, self._k), dtype=np.float32)
            check(self._L.gicp_covariances(self._h, which, _d(cov), _i(idx), _f(d2)))
            return cov, idx, d2
        check(self._L.gicp_covariances(self._h, which, _d(cov), None, None))
        return cov

    def step_correspond(self, guess=None, transformation=None):
        g = None if guess is None else _colmajor(guess)
        t = None if transformation is None else _colmajor(transformation)
        corr = np.zeros(self._n[1], dtype=np.int32)
        maha = np.zeros((self._n[1], 9), dtype=np.float32)
        m = C.c_int(0)
        check(self._L.gicp_step_correspond(self._h, None if g is None else _f(g), None if t is None else _f(t), _i(corr),
                                           _f(maha), C.byref(m)))
        return m.value, corr, maha

    def step_functor(self, mode, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        f = C.c_double(0.0)
        g = np.zeros(6)
        check(self._L.gicp_step_functor(self._h, int(mode), _d(x), C.byref(f), _d(g)))
        return f.value, g


def host_apply_state(x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    T = np.zeros(16, dtype=np.float32)
    _lib.lib().gicp_host_apply_state(_d(x), _f(T))
    return _from_colmajor(T)
