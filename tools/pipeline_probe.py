"""Per-stage wall time of a mapping-node style loop on one GPU (development aid):
raw scan -> VoxelGrid prefilter (N1) -> setInputTarget(previous) -> setInputSource(current) ->
align (node parameters: eps 0.01, 64 iterations) -> pose chaining -> global map update (N2)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt  # noqa: E402


def main():
    import torch
    torch.cuda.init()  # before the library touches the device (same order as bench.py)
    n_raw = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300000
    n_scans = 12
    world = clouds.target_surfaces(4000000, extent=120.0, n_boxes=80)
    rng = np.random.default_rng(1)
    g = ndt.NormalDistributionsTransform()
    g.setResolution(1.0)
    g.setStepSize(0.1)
    g.setTransformationEpsilon(0.01)
    g.setMaximumIterations(64)
    # scans: random subsets of the world seen from slowly moving poses
    scans = []
    for k in range(n_scans):
        T = clouds.make_T([0.25 * k, 0.05 * k, 0.0], np.deg2rad([0.0, 0.0, 0.8 * k]))
        idx = rng.choice(len(world), n_raw, replace=False)
        scans.append(clouds.apply_T(np.linalg.inv(T), world[idx] + rng.normal(0, 0.01, (n_raw, 3))))
    stages = {k: [] for k in ("filter", "target", "source", "align", "map")}
    prev = None
    pose = np.eye(4, dtype=np.float32)
    guess = np.eye(4, dtype=np.float32)
    g.mapClear()
    for k, raw in enumerate(scans):
        t0 = time.perf_counter()
        cur = g.voxelGridFilter(raw, 0.3)
        t1 = time.perf_counter()
        stages["filter"].append(t1 - t0)
        if prev is None:
            g.mapUpdate(cur, pose, 0.5)
            prev = cur
            continue
        g.setInputTarget(prev)
        t2 = time.perf_counter()
        g.setInputSource(cur)
        t3 = time.perf_counter()
        g.align(guess)
        t4 = time.perf_counter()
        T = g.getFinalTransformation() if g.hasConverged() else np.eye(4, dtype=np.float32)
        guess = T
        pose = ndt.host_chain_pose(pose, T)
        g.mapUpdate(cur, pose, 0.5)
        t5 = time.perf_counter()
        stages["target"].append(t2 - t1)
        stages["source"].append(t3 - t2)
        stages["align"].append(t4 - t3)
        stages["map"].append(t5 - t4)
        prev = cur
    host_total = sum(float(np.median(v[1:])) for v in stages.values()) * 1e3
    # ---- the same loop with every cloud resident in HBM (the *_device entry points of the C-ABI) ----
    dev_stages = {k: [] for k in ("filter", "target", "source", "align", "map")}
    raws = [torch.from_numpy(np.c_[r, np.ones(len(r), np.float32)].astype(np.float32)).cuda() for r in scans]
    bufs = [torch.empty((n_raw, 4), dtype=torch.float32, device="cuda") for _ in range(2)]
    torch.cuda.synchronize()
    g.mapClear()
    prev_n = 0
    pose = np.eye(4, dtype=np.float32)
    guess = np.eye(4, dtype=np.float32)
    for k, raw in enumerate(raws):
        cur_buf, prev_buf = bufs[k & 1], bufs[(k & 1) ^ 1]
        t0 = time.perf_counter()
        n_cur = g.voxelGridFilterDevice(raw.data_ptr(), n_raw, 16, 0.3, cur_buf.data_ptr())
        t1 = time.perf_counter()
        dev_stages["filter"].append(t1 - t0)
        if k == 0:
            g.mapUpdateDevice(cur_buf.data_ptr(), n_cur, 16, pose, 0.5)
            prev_n = n_cur
            continue
        g.setInputTargetDevice(prev_buf.data_ptr(), prev_n, 16)
        t2 = time.perf_counter()
        g.setInputSourceDevice(cur_buf.data_ptr(), n_cur, 16)
        t3 = time.perf_counter()
        g.align(guess)
        t4 = time.perf_counter()
        T = g.getFinalTransformation() if g.hasConverged() else np.eye(4, dtype=np.float32)
        guess = T
        pose = ndt.host_chain_pose(pose, T)
        g.mapUpdateDevice(cur_buf.data_ptr(), n_cur, 16, pose, 0.5)
        t5 = time.perf_counter()
        dev_stages["target"].append(t2 - t1)
        dev_stages["source"].append(t3 - t2)
        dev_stages["align"].append(t4 - t3)
        dev_stages["map"].append(t5 - t4)
        prev_n = n_cur
    print("raw scan %d pts -> %d after the 0.3 m filter; map %d pts; last registration: %d iterations, %d evaluations"
          % (n_raw, len(cur), g.mapSize(), g.getFinalNumIteration(), g.stats()["n_evals"]))
    tot = 0.0
    dtot = 0.0
    print("  stage    host clouds   clouds in HBM")
    for name, v in stages.items():
        m = float(np.median(v[1:])) * 1e3
        d = float(np.median(dev_stages[name][1:])) * 1e3
        tot += m
        dtot += d
        print("  %-7s %8.3f ms %10.3f ms" % (name, m, d))
    print("  total   %8.3f ms %10.3f ms per scan -> %.0f / %.0f scans/s" % (tot, dtot, 1e3 / tot, 1e3 / dtot))
    err = np.abs(pose[:3, 3] - [0.25 * (n_scans - 1), 0.05 * (n_scans - 1), 0.0]).max()
    print("  final pose translation error vs ground truth: %.3f m" % err)


if __name__ == "__main__":
    main()
