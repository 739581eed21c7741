"""apps/map_sequence.cpp, the node's loop one step after the other ("serial") against the overlapped pipeline (default): per-scan
wall time on a sequence of node-sized scans.   python tools/time_map_sequence.py [scans] [points per raw scan]"""
import json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toyslam_amd import _lib, clouds, ndt
_lib.build()
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_raw = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
rng = np.random.default_rng(3)
world = clouds.target_surfaces(4 * n_raw, seed=77, extent=60.0)[:, :3].astype(np.float32)
tmp = tempfile.mkdtemp(prefix="mapseq_")
pose = np.eye(4)
for k in range(1, n_scans + 1):
    pose = pose @ clouds.make_T([0.3, 0.05 * np.sin(k), 0.0], np.deg2rad([0.0, 0.0, 1.0]))
    pick = world[rng.choice(len(world), n_raw, replace=False)]
    ndt.pcd_write_xyz(os.path.join(tmp, "cloud_%d.pcd" % k), (clouds.apply_T(np.linalg.inv(pose), pick) + rng.normal(0, 0.01, pick.shape)).astype(np.float32))
exe = os.path.join(tmp, "map_sequence")
libdir = os.path.join(ROOT, "toyslam_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "apps", "map_sequence.cpp"),
                       "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
out = {}
for mode in ("serial", "pipelined_no_partition", "pipelined"):  # "pipelined": on CU partitions
    best = None
    for rep in range(3):
        env = dict(os.environ, NDT_PIPELINE_PARTITION="1" if mode == "pipelined" else "0")
        o = subprocess.check_output([exe, tmp, "0.5", "-", "node"] + ([] if mode == "serial" else ["pipeline"]), text=True, env=env)
        line = [ln for ln in o.splitlines() if ln.startswith("time:")][0]
        total = float(line.split()[2])
        kept = int([ln for ln in o.splitlines() if ln.startswith("clouds ")][0].split()[1])
        if best is None or total < best[0]:
            best = (total, line, kept)
    out[mode] = {"total_ms": best[0], "ms_per_scan": best[0] / max(1, best[2]), "scans": best[2], "line": best[1]}
print(json.dumps({"workload": "%d scans of %d raw points, 0.5 m prefilter, node parameters (apps/map_sequence.cpp)" % (n_scans, n_raw), **out}))
