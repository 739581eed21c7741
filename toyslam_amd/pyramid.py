"""Multi-resolution NDT over a streamed sequence of PCD scans (BASELINE configs[4]).

Three resident target grids (2.0 -> 1.0 -> 0.5 m voxels over the one target), one
pclomp::NormalDistributionsTransform-shaped handle per grid; every scan is registered coarse to fine, each
level's final transformation being the next level's initial guess (the `align(output, guess)` path of
ndt_rosbag_mapping_node.cpp:130).  The scans come from a directory of numbered PCD files the way the mapping node
consumes them (ndt_omp_mapping_node.cpp:110-136): the C-ABI's PcdSequence reads and parses file k+1 in a background
thread into one of two page-locked buffers while scan k is being registered, an upload thread here moves it to HBM
(H2D copy + spatial ordering, on a donor handle's own stream), and the three level handles share that one upload
(ndt_share_input_source).  Everything that computes is in libndt_mi355.so; this file only sequences calls.
"""
import os
import queue
import threading
import time

import numpy as np

from . import clouds, ndt


def write_sequence(directory, target, n_scans, n_points, max_t=0.3, max_deg=0.5, seed=clouds.SEED + 2000):
    """cloud_<k>.pcd, k = 1..n_scans: n_points target points + noise, each moved by its own T_gt,k
    (scan 0: clouds.T_GT_DEFAULT, the others U(+-max_t m, +-max_deg deg)).  Returns the T_gt list."""
    os.makedirs(directory, exist_ok=True)
    T_gts = []
    for k in range(n_scans):
        T = clouds.T_GT_DEFAULT if k == 0 else clouds.random_T(np.random.default_rng(seed + 7919 * k), max_t, max_deg)
        src = clouds.source_from_target(target, n_points, T_gt=T, seed=seed + 2 * k)
        clouds.write_pcd_xyz(os.path.join(directory, "cloud_%d.pcd" % (k + 1)), src)
        T_gts.append(T)
    return T_gts


class Pyramid:
    def __init__(self, levels=(2.0, 1.0, 0.5), device=0, trans_eps=0.01, max_iter=35, step_size=0.1, partition=None):
        """partition (default: off, NDT_PYRAMID_PARTITION=1 turns it on): the level handles register on the registration
        partition of the CUs, the donor handles upload and order the next scan on the side partition (ndt_set_cu_partition).
        Measured on this workload (round 3, NOTES.md): 272 scans/s without, 238 with -- the registration loses an eighth of
        the chip and the upload, eight times slower on 32 CUs, is still what the next scan waits for."""
        if partition is None:
            partition = os.environ.get("NDT_PYRAMID_PARTITION", "0") != "0"
        self.partition = bool(partition)
        self.resolutions = tuple(levels)
        self.levels = []
        for r in self.resolutions:
            g = ndt.NormalDistributionsTransform(device=device)
            if self.partition:
                g.setCuPartition(1)
            g.setResolution(r)
            g.setNeighborhoodSearchMethod(ndt.DIRECT7)
            g.setTransformationEpsilon(trans_eps)
            g.setMaximumIterations(max_iter)
            g.setStepSize(step_size)
            self.levels.append(g)
        self.donors = [ndt.NormalDistributionsTransform(device=device) for _ in range(2)]
        if self.partition:
            for d in self.donors:
                d.setCuPartition(2)

    def setInputTarget(self, target, is_dense=True):
        for g in self.levels:
            g.setInputTarget(target, is_dense)

    def align_donor(self, donor, guess=None):
        """Coarse-to-fine registration of the scan `donor` holds.  -> (T, per-level dicts)."""
        per = []
        T = guess
        for g in self.levels:
            t0 = time.perf_counter()
            g.shareInputSource(donor)
            g.align(T)
            T = g.getFinalTransformation()
            st = g.stats()
            per.append(dict(ms=(time.perf_counter() - t0) * 1e3, iterations=g.getFinalNumIteration(), evals=st["n_evals"],
                            hessians=st["n_hessian_recomputes"], converged=g.hasConverged(), T=T))
        return T, per

    def align(self, source, guess=None):
        self.donors[0].setInputSource(source)
        return self.align_donor(self.donors[0], guess)

    def run_sequence(self, directory, overlap=True):
        """Registers every numbered scan of `directory` against the resident grids.  overlap=True: file k+1 is read,
        parsed and uploaded while scan k is registered.  -> dict(T, seconds, per_level_ms, upload_ms, wait_ms, ...)."""
        seq = ndt.PcdSequence(directory)
        n_files = seq.poll(0)
        free = queue.Queue()
        ready = queue.Queue()
        for d in self.donors:
            free.put(d)
        upload_ms = []
        err = []

        def uploader():
            try:
                for _ in range(n_files):
                    item = seq.next_raw()
                    if item is None:
                        break
                    ptr, n, _dense, num = item
                    d = free.get()
                    t0 = time.perf_counter()
                    d.setInputSourceRaw(ptr, n, 16)  # returns when the pinned buffer is free again
                    upload_ms.append((time.perf_counter() - t0) * 1e3)
                    ready.put((d, num, n))
            except Exception as e:  # surfaced by the consumer
                err.append(e)
            ready.put(None)

        t_start = time.perf_counter()
        Ts, per_scan, wait_ms, numbers = [], [], [], []
        if not overlap:  # strictly one after the other: read / parse -> upload -> register
            for _ in range(n_files):
                t0 = time.perf_counter()
                item = seq.next_raw()
                if item is None:
                    break
                ptr, n_pts, _dense, num = item
                wait_ms.append((time.perf_counter() - t0) * 1e3)
                t0 = time.perf_counter()
                self.donors[0].setInputSourceRaw(ptr, n_pts, 16)
                upload_ms.append((time.perf_counter() - t0) * 1e3)
                T, per = self.align_donor(self.donors[0])
                Ts.append(T)
                per_scan.append(per)
                numbers.append(num)
            ready.put(None)
            wait_ms.append(0.0)
        else:
            th = threading.Thread(target=uploader, daemon=True)
            th.start()
        while overlap or not ready.empty():
            t0 = time.perf_counter()
            item = ready.get()
            wait_ms.append((time.perf_counter() - t0) * 1e3)
            if item is None:
                break
            d, num, _n = item
            T, per = self.align_donor(d)
            free.put(d)
            Ts.append(T)
            per_scan.append(per)
            numbers.append(num)
        seconds = time.perf_counter() - t_start
        if overlap:
            th.join()
        if err:
            raise err[0]
        n = max(1, len(Ts))
        per_level_ms = [float(np.mean([p[i]["ms"] for p in per_scan])) if per_scan else 0.0 for i in range(len(self.levels))]
        return dict(T=Ts, file_numbers=numbers, seconds=seconds, per_level_ms=per_level_ms, per_scan=per_scan,
                    upload_ms=float(np.mean(upload_ms)) if upload_ms else 0.0,
                    wait_ms=float(np.sum(wait_ms[:-1]) / n) if wait_ms else 0.0,
                    evals_per_scan=float(np.mean([sum(l["evals"] + l["hessians"] for l in p) for p in per_scan])) if per_scan else 0.0)
