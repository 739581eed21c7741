"""Throughput of the PCD read-ahead alone (no GPU work): files per second of ndt_pcd_sequence_next over a directory of
2M-point binary scans.   python tools/time_pcd_sequence.py [n_files] [points]"""
import json, os, sys, tempfile, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt

n_files = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n_pts = int(float(sys.argv[2])) if len(sys.argv) > 2 else 2000000
d = tempfile.mkdtemp(prefix="pcdseq_")
rng = np.random.default_rng(1)
pts = rng.normal(0, 30, (n_pts, 3)).astype(np.float32)
for k in range(n_files):
    clouds.write_pcd_xyz(os.path.join(d, "cloud_%d.pcd" % (k + 1)), pts)
for trial in range(2):
    seq = ndt.PcdSequence(d)
    n = seq.poll(0)
    t0 = time.perf_counter()
    got = 0
    while True:
        item = seq.next_raw()
        if item is None:
            break
        got += 1
    dt = time.perf_counter() - t0
    print(json.dumps({"files": got, "points_per_file": n_pts, "ms_per_file": round(dt / got * 1e3, 2), "GBs": round(got * n_pts * 12 / dt / 1e9, 2)}))
for f in os.listdir(d):
    os.remove(os.path.join(d, f))
os.rmdir(d)
