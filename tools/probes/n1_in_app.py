"""NDT_TRACE_N1 inside apps/map_sequence (development aid): where the prefilter's time goes in the node loop."""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from toyslam_amd import _lib, clouds, ndt
n_scans, n_raw = 40, 60000
rng = np.random.default_rng(3)
world = clouds.target_surfaces(4 * n_raw, seed=77, extent=60.0)[:, :3].astype(np.float32)
tmp = tempfile.mkdtemp(prefix="nodeloop_")
pose = np.eye(4)
for k in range(1, n_scans + 1):
    pose = pose @ clouds.make_T([0.3, 0.05 * np.sin(k), 0.0], np.deg2rad([0.0, 0.0, 1.0]))
    pick = world[rng.choice(len(world), n_raw, replace=False)]
    ndt.pcd_write_xyz(os.path.join(tmp, "cloud_%d.pcd" % k), (clouds.apply_T(np.linalg.inv(pose), pick) + rng.normal(0, 0.01, pick.shape)).astype(np.float32))
exe = os.path.join(tmp, "map_sequence")
libdir = os.path.join(ROOT, "toyslam_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "apps", "map_sequence.cpp"),
                       "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
for zc in ("1", "0", "1", "0", "1", "0"):
    r = subprocess.run([exe, tmp, "0.5", "-", "node"], text=True, capture_output=True, env=dict(os.environ, MAP_SEQUENCE_OVERLAP=zc))
    print("MAP_SEQUENCE_OVERLAP=" + zc, [ln for ln in r.stdout.splitlines() if ln.startswith("time:")][-1])
