"""Python host-side mirror of pclomp::NormalDistributionsTransform over the C-ABI.

Method names follow the reference class (ndt_omp/include/pclomp/ndt_omp.h:70-502
and the pcl::Registration methods its callers use) so that tests read like the
reference's call sites (ndt_omp/apps/align.cpp:14-33,
lidar_subscriber/src/ndt_omp_mapping_node.cpp:151-169).  All compute happens in
libndt_mi355.so on the GPU; nothing here falls back to numpy.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import COMM_ID_BYTES, DIRECT1, DIRECT7, DIRECT26, KDTREE, NdtError, check  # noqa: F401


def _f(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _i(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _cloud(a):
    """(N, >=3) float32 C-contiguous view; stride = row bytes (16 for XYZ+pad, 32 for XYZI...)."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] < 3:
        raise ValueError("cloud must be (N, >=3)")
    return a


def _colmajor(T):
    return np.ascontiguousarray(np.asarray(T, dtype=np.float32).T).reshape(16)


def _from_colmajor(v):
    return np.asarray(v, dtype=np.float32).reshape(4, 4).T.copy()


class DeviceCloud:
    """An ndt_cloud: a cloud resident in HBM with its bounding boxes (include/ndt_mi355.h, "clouds that stay in HBM")."""

    def __init__(self, owner, c):
        self._owner, self._c, self._L = owner, c, owner._L

    def __len__(self):
        n = C.c_size_t(0)
        check(self._L.ndt_cloud_size(self._c, C.byref(n)))
        return n.value

    def data_ptr(self):
        p, n = C.c_void_p(None), C.c_size_t(0)
        check(self._L.ndt_cloud_data(self._c, C.byref(p), C.byref(n)))
        return p.value or 0

    def numpy(self):
        n = len(self)
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        check(self._L.ndt_cloud_download(self._owner._h, self._c, out.ctypes.data, 16))
        return out[:n, :3].copy()

    def release(self):
        if self._c:
            self._L.ndt_cloud_release(self._c)
            self._c = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


class NormalDistributionsTransform:
    """Drop-in shaped like pclomp::NormalDistributionsTransform<PointT, PointT>."""

    def __init__(self, device=0, _handle=None):
        self._L = _lib.lib()
        self._device = device
        if _handle is None:
            h = C.c_void_p()
            check(self._L.ndt_create(device, C.byref(h)))
            self._h = h
        else:
            self._h = _handle
        self._keep = []  # keeps callback objects alive

    def __del__(self):
        try:
            self._L.ndt_destroy(self._h)
        except Exception:
            pass

    def copy(self):
        """Copy-construction (the nodes return the object by value): shares the device grid."""
        h = C.c_void_p()
        check(self._L.ndt_clone(self._h, C.byref(h)))
        return NormalDistributionsTransform(_handle=h)

    # ---- pclomp setters / getters (ndt_omp.h:115-209) -------------------------
    def setNumThreads(self, n):
        check(self._L.ndt_set_num_threads(self._h, int(n)))

    def setResolution(self, r):
        check(self._L.ndt_set_resolution(self._h, float(r)))

    def getResolution(self):
        return self._L.ndt_get_resolution(self._h)

    def setStepSize(self, s):
        check(self._L.ndt_set_step_size(self._h, float(s)))

    def getStepSize(self):
        return self._L.ndt_get_step_size(self._h)

    def setOutlierRatio(self, r):
        check(self._L.ndt_set_outlier_ratio(self._h, float(r)))

    def getOutlierRatio(self):
        return self._L.ndt_get_outlier_ratio(self._h)

    def setNeighborhoodSearchMethod(self, m):
        check(self._L.ndt_set_neighborhood_search_method(self._h, int(m)))

    def setTransformationEpsilon(self, e):
        check(self._L.ndt_set_transformation_epsilon(self._h, float(e)))

    def setMaximumIterations(self, n):
        check(self._L.ndt_set_maximum_iterations(self._h, int(n)))

    def setMinPointPerVoxel(self, n):
        check(self._L.ndt_set_min_points_per_voxel(self._h, int(n)))

    def setCovEigValueInflationRatio(self, r):
        check(self._L.ndt_set_cov_eig_value_inflation_ratio(self._h, float(r)))

    # ---- inputs -------------------------------------------------------------------
    def setInputTarget(self, cloud, is_dense=True):
        a = _cloud(cloud)
        check(self._L.ndt_set_input_target(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, int(is_dense)))

    def setInputSource(self, cloud):
        a = _cloud(cloud)
        check(self._L.ndt_set_input_source(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4))

    def setInputTargetDevice(self, dev_ptr, n, stride_bytes, is_dense=True):
        check(self._L.ndt_set_input_target_device(self._h, C.c_void_p(dev_ptr), n, stride_bytes, int(is_dense)))

    def setInputTargetDeviceRef(self, dev_ptr, n, is_dense=True):
        """The cloud (n 16-byte records in HBM) is used where it lies: keep it alive and unchanged while it is the target."""
        check(self._L.ndt_set_input_target_device_ref(self._h, C.c_void_p(dev_ptr), n, int(is_dense)))

    def setInputSourceDeviceRef(self, dev_ptr, n):
        check(self._L.ndt_set_input_source_device_ref(self._h, C.c_void_p(dev_ptr), n))

    def setInputSourceDevice(self, dev_ptr, n, stride_bytes):
        check(self._L.ndt_set_input_source_device(self._h, C.c_void_p(dev_ptr), n, stride_bytes))

    # ---- registration -------------------------------------------------------------
    def align(self, guess=None, n_out=None):
        """align(output[, guess]).  Returns the aligned cloud (N,4) when n_out is given, else None."""
        g = None if guess is None else _colmajor(guess)
        out = np.zeros((n_out, 4), dtype=np.float32) if n_out else None
        check(self._L.ndt_align(self._h, _f(g) if g is not None else None, None, None, None, None,
                                out.ctypes.data if out is not None else None, 16))
        return out

    def _result(self):
        T = np.zeros(16, dtype=np.float32)
        conv, it = C.c_int(0), C.c_int(0)
        tp = C.c_double(0)
        check(self._L.ndt_get_result(self._h, _f(T), C.byref(conv), C.byref(it), C.byref(tp)))
        return _from_colmajor(T), bool(conv.value), it.value, tp.value

    def hasConverged(self):
        return self._result()[1]

    def getFinalTransformation(self):
        return self._result()[0]

    def getFinalNumIteration(self):
        return self._result()[2]

    def getTransformationProbability(self):
        return self._result()[3]

    def stats(self):
        ne, nh = C.c_int(0), C.c_int(0)
        nn = C.c_double(0)
        check(self._L.ndt_get_stats(self._h, C.byref(ne), C.byref(nh), C.byref(nn)))
        return dict(n_evals=ne.value, n_hessian_recomputes=nh.value, mean_neighbors=nn.value)

    def output_device(self):
        p = C.c_void_p()
        n = C.c_size_t(0)
        check(self._L.ndt_get_output_device(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def calculateScore(self, cloud):
        a = _cloud(cloud)
        s = C.c_double(0)
        check(self._L.ndt_calculate_score(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, C.byref(s)))
        return s.value

    # ---- scan prefilter (pcl::VoxelGrid) --------------------------------------------
    def voxelGridFilter(self, cloud, leaf_size, is_dense=True):
        """pcl::VoxelGrid::filter on the GPU: (V, 3) float32 centroids in ascending voxel-index order.
        Raises NdtError(GRID_OVERFLOW) where PCL warns and passes the input through."""
        a = _cloud(cloud)
        out = np.zeros((max(a.shape[0], 1), 4), dtype=np.float32)
        n = C.c_size_t(0)
        check(self._L.ndt_voxel_grid_filter(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, int(is_dense),
                                            float(leaf_size), out.ctypes.data, 16, C.byref(n)))
        return out[:n.value, :3].copy()

    def voxelGridFilterDevice(self, dev_ptr, n, stride_bytes, leaf_size, out_dev_ptr, is_dense=True):
        m = C.c_size_t(0)
        check(self._L.ndt_voxel_grid_filter_device(self._h, C.c_void_p(dev_ptr), n, stride_bytes, int(is_dense),
                                                   float(leaf_size), C.c_void_p(out_dev_ptr), C.byref(m)))
        return m.value

    def getFitnessScore(self, max_range=np.finfo(np.float64).max):
        """pcl::Registration::getFitnessScore of the last align (exact NN search on the GPU)."""
        v = C.c_double(0)
        check(self._L.ndt_get_fitness_score(self._h, float(max_range), C.byref(v)))
        return v.value

    # ---- global map (N2) ------------------------------------------------------------
    def mapClear(self):
        check(self._L.ndt_map_clear(self._h))

    def mapUpdate(self, scan, pose=None, leaf_size=0.5, is_dense=True):
        """update_global_map of the mapping nodes: transform the scan by `pose`, append it to the
        HBM-resident map, voxel-filter the map.  Returns (map size, overflowed)."""
        a = _cloud(scan)
        T = None if pose is None else _colmajor(pose)
        ov = C.c_int(0)
        check(self._L.ndt_map_update(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, int(is_dense),
                                     _f(T) if T is not None else None, float(leaf_size), C.byref(ov)))
        return self.mapSize(), bool(ov.value)

    def mapUpdateDevice(self, dev_ptr, n, stride_bytes, pose=None, leaf_size=0.5, is_dense=True):
        T = None if pose is None else _colmajor(pose)
        ov = C.c_int(0)
        check(self._L.ndt_map_update_device(self._h, C.c_void_p(dev_ptr), n, stride_bytes, int(is_dense),
                                            _f(T) if T is not None else None, float(leaf_size), C.byref(ov)))
        return self.mapSize(), bool(ov.value)

    def mapSize(self):
        n = C.c_size_t(0)
        check(self._L.ndt_map_size(self._h, C.byref(n)))
        return n.value

    def mapGet(self):
        n = self.mapSize()
        out = np.zeros((max(n, 1), 4), dtype=np.float32)
        check(self._L.ndt_map_get(self._h, out.ctypes.data, 16))
        return out[:n, :3].copy()

    # ---- clouds that stay in HBM (ndt_cloud) ------------------------------------------
    def voxelGridFilterCloud(self, cloud, leaf_size, is_dense=True):
        """N1 with the result left in HBM: -> (DeviceCloud, overflowed)."""
        a = _cloud(cloud)
        c, ov = C.c_void_p(None), C.c_int(0)
        check(self._L.ndt_cloud_voxel_filter(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, int(is_dense), float(leaf_size), 0,
                                             C.byref(c), C.byref(ov)))
        return DeviceCloud(self, c), bool(ov.value)

    def voxelGridFilterCloudDevice(self, dev_ptr, n, stride_bytes, leaf_size, is_dense=True):
        """The same with the input already in HBM."""
        c, ov = C.c_void_p(None), C.c_int(0)
        check(self._L.ndt_cloud_voxel_filter(self._h, C.c_void_p(dev_ptr), n, stride_bytes, int(is_dense), float(leaf_size), 1,
                                             C.byref(c), C.byref(ov)))
        return DeviceCloud(self, c), bool(ov.value)

    def voxelGridFilterBegin(self, dc, leaf_size, is_dense=True):
        """N1 of an ndt_cloud, first half: queued on the handle's filter stream, not waited for."""
        check(self._L.ndt_cloud_voxel_filter_begin(self._h, dc._c, int(is_dense), float(leaf_size)))

    def voxelGridFilterEnd(self):
        c, ov = C.c_void_p(None), C.c_int(0)
        check(self._L.ndt_cloud_voxel_filter_end(self._h, C.byref(c), C.byref(ov)))
        return DeviceCloud(self, c), bool(ov.value)

    def warmUp(self, expected_scan_points=0):
        check(self._L.ndt_warm_up(self._h, int(expected_scan_points)))

    def uploadCloud(self, cloud):
        a = _cloud(cloud)
        c = C.c_void_p(None)
        check(self._L.ndt_cloud_upload(self._h, a.ctypes.data, a.shape[0], a.shape[1] * 4, C.byref(c)))
        return DeviceCloud(self, c)

    def setInputSourceCloud(self, dc):
        check(self._L.ndt_set_input_source_cloud(self._h, dc._c))

    def setInputTargetCloud(self, dc, is_dense=True):
        check(self._L.ndt_set_input_target_cloud(self._h, dc._c, int(is_dense)))

    def promoteSourceToTarget(self, is_dense=True):
        """The current input source becomes the input target (cloud k of pair (k-1, k) is the target of pair (k, k+1))."""
        check(self._L.ndt_promote_source_to_target(self._h, int(is_dense)))

    def mapUpdateCloud(self, dc, pose=None, leaf_size=0.5, is_dense=True):
        T = None if pose is None else _colmajor(pose)
        ov = C.c_int(0)
        check(self._L.ndt_map_update_cloud(self._h, dc._c, int(is_dense), _f(T) if T is not None else None, float(leaf_size), C.byref(ov)))
        return self.mapSize(), bool(ov.value)

    # ---- batch (map-build) ---------------------------------------------------------
    def alignBatch(self, clouds=None, guesses=None, device_ptr=None, offsets=None, stride_bytes=16):
        """Register many sources against the one target in lock-step.

        clouds: list of (N_k, >=3) arrays (host), or device_ptr + offsets for HBM-resident data."""
        if clouds is not None:
            cols = {c.shape[1] for c in clouds}
            if len(cols) != 1:
                raise ValueError("all clouds must have the same column count")
            cat = _cloud(np.concatenate(clouds, axis=0))
            offsets = np.zeros(len(clouds) + 1, dtype=np.uintp)
            offsets[1:] = np.cumsum([c.shape[0] for c in clouds])
            ptr, stride, fn = cat.ctypes.data, cat.shape[1] * 4, self._L.ndt_align_batch
        else:
            offsets = np.ascontiguousarray(offsets, dtype=np.uintp)
            ptr, stride, fn = C.c_void_p(device_ptr), stride_bytes, self._L.ndt_align_batch_device
        B = len(offsets) - 1
        g = None
        if guesses is not None:
            g = np.ascontiguousarray(np.stack([_colmajor(x) for x in guesses]))
        T = np.zeros((B, 16), dtype=np.float32)
        conv = np.zeros(B, dtype=np.int32)
        it = np.zeros(B, dtype=np.int32)
        tp = np.zeros(B, dtype=np.float64)
        check(fn(self._h, ptr, offsets.ctypes.data_as(C.POINTER(C.c_size_t)), B, stride,
                 _f(g) if g is not None else None, _f(T), _i(conv), _i(it), _d(tp)))
        return dict(T=np.stack([_from_colmajor(T[k]) for k in range(B)]), converged=conv.astype(bool),
                    iterations=it, trans_probability=tp)

    def alignBatchSharded(self, clouds=None, first_scan=0, total_scans=None, guesses=None, device_ptr=None, offsets=None,
                          stride_bytes=16):
        """The lock-step batch with the scans sharded over ranks (ndt_align_batch_sharded*): this rank holds scans
        [first_scan, first_scan + n_local) of total_scans; needs a communicator (commInitRank) or an all-reduce hook.
        Outputs cover all total_scans scans and are identical on every rank."""
        if clouds is not None:
            cat = _cloud(np.concatenate(clouds, axis=0)) if clouds else np.zeros((0, 4), np.float32)
            offsets = np.zeros(len(clouds) + 1, dtype=np.uintp)
            offsets[1:] = np.cumsum([c.shape[0] for c in clouds])
            ptr, stride, fn = cat.ctypes.data, cat.shape[1] * 4, self._L.ndt_align_batch_sharded
        else:
            offsets = np.ascontiguousarray(offsets, dtype=np.uintp)
            ptr, stride, fn = C.c_void_p(device_ptr), stride_bytes, self._L.ndt_align_batch_sharded_device
        n_local = len(offsets) - 1
        B = int(total_scans if total_scans is not None else n_local)
        g = None
        if guesses is not None:
            g = np.ascontiguousarray(np.stack([_colmajor(x) for x in guesses]))
            assert g.shape[0] == B
        T = np.zeros((B, 16), dtype=np.float32)
        conv = np.zeros(B, dtype=np.int32)
        it = np.zeros(B, dtype=np.int32)
        tp = np.zeros(B, dtype=np.float64)
        check(fn(self._h, ptr, offsets.ctypes.data_as(C.POINTER(C.c_size_t)), n_local, int(first_scan), B, stride,
                 _f(g) if g is not None else None, _f(T), _i(conv), _i(it), _d(tp)))
        return dict(T=np.stack([_from_colmajor(T[k]) for k in range(B)]), converged=conv.astype(bool),
                    iterations=it, trans_probability=tp)

    # ---- multi-GPU: native RCCL communicator (ndt_comm_*) -----------------------------
    def commInitRank(self, unique_id, rank, world_size):
        """unique_id: the COMM_ID_BYTES bytes rank 0 got from comm_get_unique_id()."""
        buf = (C.c_char * COMM_ID_BYTES).from_buffer_copy(bytes(unique_id))
        check(self._L.ndt_comm_init_rank(self._h, C.cast(buf, C.c_void_p), int(rank), int(world_size)))

    def commDestroy(self):
        check(self._L.ndt_comm_destroy(self._h))

    def commStats(self):
        r, w, ls = C.c_int(0), C.c_int(0), C.c_int(0)
        n = C.c_longlong(0)
        check(self._L.ndt_comm_stats(self._h, C.byref(r), C.byref(w), C.byref(n), C.byref(ls)))
        return dict(rank=r.value, world=w.value, collectives=n.value, lock_steps=ls.value)

    def shareInputSource(self, donor):
        """Register the source cloud `donor` has uploaded (ndt_share_input_source)."""
        check(self._L.ndt_share_input_source(self._h, donor._h))

    def shareInputTarget(self, donor):
        """Take the target cloud and the voxel grid `donor` built (no copy, no rebuild)."""
        check(self._L.ndt_share_input_target(self._h, donor._h))

    def setCuPartition(self, partition):
        """0 whole device, 1 registration partition, 2 side partition (ndt_set_cu_partition)."""
        check(self._L.ndt_set_cu_partition(self._h, int(partition)))

    def getCuPartition(self):
        a, b = C.c_int(0), C.c_int(0)
        check(self._L.ndt_get_cu_partition(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def setInputSourceRaw(self, host_ptr, n, stride_bytes=16):
        """setInputSource from a raw host pointer (e.g. the page-locked buffer of a PcdSequence scan)."""
        check(self._L.ndt_set_input_source(self._h, C.c_void_p(host_ptr), n, stride_bytes))

    def setVoxelIndex(self, mode):
        """0 = dense / sparse voxel index chosen by occupancy, 1 = dense table, 2 = sparse (sorted build + hash look-up)."""
        check(self._L.ndt_set_voxel_index(self._h, int(mode)))

    def setBatchGroups(self, n):
        """Independent lock-step groups alignBatch runs as (0 = automatic, 1 = one loop)."""
        check(self._L.ndt_set_batch_groups(self._h, int(n)))

    def setAllreduce(self, fn, on_device=False):
        """fn(buffer_address, n_doubles, on_device) -> 0 on success; None removes the hook."""
        if fn is None:
            cb = _lib.ALLREDUCE_FN(0)
        else:
            def tramp(buf, n, dev, _user):
                try:
                    return int(fn(buf, n, bool(dev)) or 0)
                except Exception:  # never unwind through C
                    import traceback
                    traceback.print_exc()
                    return 1
            cb = _lib.ALLREDUCE_FN(tramp)
        self._keep = [cb]
        check(self._L.ndt_set_allreduce(self._h, cb, None, int(on_device)))

    def diag_stamps(self, p, max_waves=1 << 16):
        p = np.ascontiguousarray(p, dtype=np.float64)
        st = np.zeros((max_waves, 8), dtype=np.uint64)
        n = C.c_size_t(max_waves)
        check(self._L.ndt_diag_stamps(self._h, _d(p), st.ctypes.data_as(C.POINTER(C.c_ulonglong)), C.byref(n)))
        return st[:n.value]

    def diag_server_roundtrip(self, p, n_iter=200):
        p = np.ascontiguousarray(p, dtype=np.float64)
        us = np.zeros(3)
        check(self._L.ndt_diag_server_roundtrip(self._h, _d(p), n_iter, _d(us)))
        return dict(nop_us=us[0], no_hessian_us=us[1], with_hessian_us=us[2])

    def diag_selfdrive(self, p, rounds=200):
        """Rounds driven from the device (no host in the loop, no solver step): us per round without / with the body."""
        p = np.ascontiguousarray(p, dtype=np.float64)
        us = np.zeros(2)
        check(self._L.ndt_diag_selfdrive(self._h, _d(p), rounds, _d(us)))
        return dict(protocol_only_us=us[0], with_hessian_body_us=us[1])

    def selftest_reduce(self, n_blocks=3):
        out = np.zeros((n_blocks, _lib.EVAL_STRIDE))
        check(self._L.ndt_selftest_reduce(self._h, n_blocks, _d(out)))
        return out

    def selftest_server_idle(self, p, stall_ms):
        p = np.ascontiguousarray(p, dtype=np.float64)
        served = C.c_int(-1)
        scores = np.zeros(3)
        check(self._L.ndt_selftest_server_idle(self._h, _d(p), int(stall_ms), C.byref(served), _d(scores)))
        return bool(served.value), scores

    def setEvaluationPath(self, persistent):
        """True (default): one persistent kernel per registration; False: one launch per evaluation."""
        check(self._L.ndt_set_evaluation_path(self._h, int(bool(persistent))))

    def profile(self, on):
        check(self._L.ndt_profile_enable(self._h, int(on)))

    def profile_read(self, kind=0, reset=True):
        """(launches, total_ms) of the kernel of `kind` (0/1/2: per-evaluation launches of profile(1);
        3: the per-registration persistent kernel of profile(2)), from HIP events on the handle's stream."""
        n = C.c_longlong(0)
        ms = C.c_double(0)
        check(self._L.ndt_profile_read(self._h, kind, C.byref(n), C.byref(ms), int(reset)))
        return n.value, ms.value

    # ---- inspection ------------------------------------------------------------------
    def eval(self, p, compute_hessian=True, T=None):
        p = np.ascontiguousarray(p, dtype=np.float64)
        score, nn = C.c_double(0), C.c_double(0)
        g = np.zeros(6)
        H = np.zeros(36) if compute_hessian else None
        if T is None:
            check(self._L.ndt_eval(self._h, _d(p), C.byref(score), _d(g), _d(H) if compute_hessian else None,
                                   C.byref(nn)))
        else:
            Tc = _colmajor(T)
            check(self._L.ndt_eval_with_matrix(self._h, _f(Tc), _d(p), C.byref(score), _d(g),
                                               _d(H) if compute_hessian else None, C.byref(nn)))
        return score.value, g, (H.reshape(6, 6) if compute_hessian else None), nn.value

    def hessian_f64(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        H = np.zeros(36)
        check(self._L.ndt_eval_hessian_f64(self._h, _d(p), _d(H)))
        return H.reshape(6, 6)

    def grid_counts(self):
        """Occupied / valid voxel counts of the target grid."""
        nl, nv = C.c_size_t(0), C.c_size_t(0)
        check(self._L.ndt_grid_size(self._h, C.byref(nl), C.byref(nv)))
        return dict(n_leaves=nl.value, n_valid=nv.value)

    def grid(self):
        nl, nv = C.c_size_t(0), C.c_size_t(0)
        check(self._L.ndt_grid_size(self._h, C.byref(nl), C.byref(nv)))
        n = nl.value
        idx = np.zeros(n, dtype=np.int64)
        npts = np.zeros(n, dtype=np.int32)
        mean = np.zeros((n, 3))
        cov = np.zeros((n, 3, 3))
        icov = np.zeros((n, 3, 3))
        evals = np.zeros((n, 3))
        if n:
            check(self._L.ndt_grid_dump(self._h, idx.ctypes.data_as(C.POINTER(C.c_int64)), _i(npts), _d(mean), _d(cov),
                                        _d(icov), _d(evals)))
        mb, xb, db = (np.zeros(3, dtype=np.int32) for _ in range(3))
        check(self._L.ndt_grid_info(self._h, _i(mb), _i(xb), _i(db)))
        return dict(idx=idx, n=npts, mean=mean, cov=cov, icov=icov, evals=evals, min_b=mb, max_b=xb, div_b=db,
                    n_valid=nv.value)


def comm_get_unique_id():
    """ncclGetUniqueId through the C-ABI (rank 0): COMM_ID_BYTES bytes to carry to every rank."""
    buf = (C.c_char * COMM_ID_BYTES)()
    check(_lib.lib().ndt_comm_get_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf.raw)


# ---- host-only scalar pieces (no GPU needed) ---------------------------------------
def host_solve6(H, b):
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(36)
    b = np.ascontiguousarray(b, dtype=np.float64)
    x = np.zeros(6)
    _lib.lib().ndt_host_solve6(_d(H), _d(b), _d(x))
    return x


def pcd_read_xyz(path):
    """loadPCDFile<PointXYZ> through the C-ABI: ((N, 3) float32, is_dense)."""
    L = _lib.lib()
    n, nf, kind = C.c_size_t(0), C.c_int(0), C.c_int(0)
    check(L.ndt_pcd_read_header(os.fsencode(path), C.byref(n), C.byref(nf), C.byref(kind)))
    out = np.zeros((max(n.value, 1), 4), dtype=np.float32)
    dense = C.c_int(1)
    check(L.ndt_pcd_read_xyz(os.fsencode(path), out.ctypes.data, n.value, 16, C.byref(n), C.byref(dense)))
    return out[:n.value, :3].copy(), bool(dense.value)


def pcd_write_xyz(path, xyz, binary=True):
    a = _cloud(xyz)
    check(_lib.lib().ndt_pcd_write_xyz(os.fsencode(path), a.ctypes.data, a.shape[0], a.shape[1] * 4, int(binary)))


def host_chain_pose(pose, transform):
    """pose * transform in Eigen's f32 rounding (the nodes' trajectory chaining)."""
    out = np.zeros(16, dtype=np.float32)
    _lib.lib().ndt_host_chain_pose(_f(_colmajor(pose)), _f(_colmajor(transform)), _f(out))
    return _from_colmajor(out)


def host_pose_to_matrix(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    T = np.zeros(16, dtype=np.float32)
    _lib.lib().ndt_host_pose_to_matrix(_d(p), _f(T))
    return _from_colmajor(T)


def host_matrix_to_pose(T):
    Tc = _colmajor(T)
    p = np.zeros(6)
    _lib.lib().ndt_host_matrix_to_pose(_f(Tc), _d(p))
    return p


def host_angle_derivatives(p):
    p = np.ascontiguousarray(p, dtype=np.float64)
    j = np.zeros((8, 3), dtype=np.float32)
    h = np.zeros((15, 3), dtype=np.float32)
    jd = np.zeros((8, 3))
    hd = np.zeros((15, 3))
    _lib.lib().ndt_host_angle_derivatives(_d(p), _f(j), _f(h), _d(jd), _d(hd))
    return j, h, jd, hd


def host_thread_budget():
    """(CPUs in the affinity mask, cgroup CPU bandwidth in CPUs or 0.0 for unlimited, LOCAL_WORLD_SIZE) as the library probes them."""
    a, w, q = C.c_int(0), C.c_int(0), C.c_double(0)
    _lib.lib().ndt_host_thread_budget(C.byref(a), C.byref(q), C.byref(w))
    return a.value, q.value, w.value


def host_thread_plan(affinity_cpus, quota_cpus, local_world_size):
    """(pool threads, max batch groups) the library would use for that budget (pure)."""
    p, g = C.c_int(0), C.c_int(0)
    _lib.lib().ndt_host_thread_plan(int(affinity_cpus), float(quota_cpus), int(local_world_size), C.byref(p), C.byref(g))
    return p.value, g.value


def host_gauss(resolution, outlier_ratio):
    d = np.zeros(3)
    _lib.lib().ndt_host_gauss(float(resolution), float(outlier_ratio), _d(d))
    return d


def host_run_driver(evaluator, n_source, guess=None, resolution=1.0, step_size=0.1, outlier_ratio=0.55,
                    trans_eps=0.1, max_iter=35):
    """Run the PRODUCT Newton/More-Thuente driver against a Python evaluator
    evaluator(kind, T(4x4), p(6)) -> (score, g(6), H(6x6))."""
    def tramp(_user, kind, Tp, pp, score, g, H):
        try:
            T = _from_colmajor(np.ctypeslib.as_array(Tp, shape=(16,)))
            p = np.ctypeslib.as_array(pp, shape=(6,)).copy()
            s, gg, HH = evaluator(kind, T, p)
            score[0] = s
            for k in range(6):
                g[k] = gg[k]
            HH = np.asarray(HH, dtype=np.float64).reshape(36)
            for k in range(36):
                H[k] = HH[k]
            return 0
        except Exception:
            import traceback
            traceback.print_exc()
            return 1
    cb = _lib.EVAL_CB(tramp)
    g = None if guess is None else _colmajor(guess)
    T = np.zeros(16, dtype=np.float32)
    conv, it, ne, nh = (C.c_int(0) for _ in range(4))
    tp = C.c_double(0)
    check(_lib.lib().ndt_host_run_driver(cb, None, n_source, _f(g) if g is not None else None, resolution, step_size,
                                         outlier_ratio, trans_eps, max_iter, _f(T), C.byref(conv), C.byref(it),
                                         C.byref(tp), C.byref(ne), C.byref(nh)))
    return dict(T=_from_colmajor(T), converged=bool(conv.value), iterations=it.value, trans_probability=tp.value,
                n_evals=ne.value, n_hessian_recomputes=nh.value)


def extract_file_number(stem):
    """extract_file_number of the mapping node (ndt_omp_mapping_node.cpp:231-239)."""
    return _lib.lib().ndt_host_extract_file_number(stem.encode())


class PcdSequence:
    """The numbered *.pcd scans of a directory in the mapping node's order (process_new_clouds,
    ndt_omp_mapping_node.cpp:110-136), the next file being read in the background."""

    def __init__(self, directory):
        self._L = _lib.lib()
        h = C.c_void_p()
        check(self._L.ndt_pcd_sequence_open(os.fsencode(directory), C.byref(h)))
        self._h = h

    def __del__(self):
        try:
            self._L.ndt_pcd_sequence_close(self._h)
        except Exception:
            pass

    def poll(self, loaded_clouds):
        n = C.c_size_t(0)
        check(self._L.ndt_pcd_sequence_poll(self._h, loaded_clouds, C.byref(n)))
        return n.value

    def next(self):
        """-> (xyz (n,3) float32 copy, is_dense, file_number) or None when nothing is queued."""
        p, n, dense, num = C.c_void_p(), C.c_size_t(0), C.c_int(1), C.c_int(-1)
        check(self._L.ndt_pcd_sequence_next(self._h, C.byref(p), C.byref(n), C.byref(dense), C.byref(num)))
        if not p.value:
            return None
        a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 4))
        return a[:, :3].copy(), bool(dense.value), num.value

    def stage(self, device=0):
        """From now on the reading threads copy every scan to `device` as soon as its file is parsed (next_device)."""
        check(self._L.ndt_pcd_sequence_stage(self._h, int(device)))

    def next_device(self):
        """-> (device address, host address, n, is_dense, file_number) of the next staged scan (16-byte records, valid until
        the following call of a next* method), or None when nothing is queued."""
        d, p, n, dense, num = C.c_void_p(), C.c_void_p(), C.c_size_t(0), C.c_int(1), C.c_int(-1)
        check(self._L.ndt_pcd_sequence_next_device(self._h, C.byref(d), C.byref(p), C.byref(n), C.byref(dense), C.byref(num)))
        if not p.value:
            return None
        return d.value, p.value, n.value, bool(dense.value), num.value

    def next_raw(self):
        """-> (host address of n x (x, y, z, 1.0f) records, n, is_dense, file_number) or None.  The records sit in one of
        the sequence's two (page-locked) buffers and stay valid until the following call of next / next_raw."""
        p, n, dense, num = C.c_void_p(), C.c_size_t(0), C.c_int(1), C.c_int(-1)
        check(self._L.ndt_pcd_sequence_next(self._h, C.byref(p), C.byref(n), C.byref(dense), C.byref(num)))
        if not p.value:
            return None
        return p.value, n.value, bool(dense.value), num.value


def repack_fields(data, n, point_step, off_x=0, off_y=4, off_z=8):
    """PointCloud2-style records (bytes-like, any step / offsets) -> ((n, 4) float32 x,y,z,1, is_dense)."""
    buf = np.frombuffer(data, dtype=np.uint8)
    assert buf.size >= n * point_step
    out = np.zeros((n, 4), dtype=np.float32)
    dense = C.c_int(1)
    check(_lib.lib().ndt_host_repack_fields(buf.ctypes.data, n, point_step, off_x, off_y, off_z, out.ctypes.data, C.byref(dense)))
    return out, bool(dense.value)
