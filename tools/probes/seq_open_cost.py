"""What a pass over a PCD sequence pays before its first scan is there: open, poll, first next (development aid)."""
import sys, time, os, numpy as np, tempfile, shutil
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toyslam_amd import clouds, ndt
n = 2000000
p = tempfile.mkdtemp(prefix="seqopen_")
rng = np.random.default_rng(1)
for k in range(8):
    clouds.write_pcd_xyz("%s/cloud_%d.pcd" % (p, k + 1), rng.standard_normal((n, 3)).astype(np.float32))
g = ndt.NormalDistributionsTransform()  # (a device context)
for rnd in range(4):
    t0 = time.perf_counter(); s = ndt.PcdSequence(p); t1 = time.perf_counter(); m = s.poll(0); t2 = time.perf_counter()
    a = s.next_raw(); t3 = time.perf_counter()
    rest = []
    while True:
        t = time.perf_counter(); a = s.next_raw(); rest.append((time.perf_counter() - t) * 1e3)
        if a is None: break
    t4 = time.perf_counter(); del s; t5 = time.perf_counter()
    print("open %.2f ms | poll (%d files) %.2f | first next %.2f | following nexts %s | close %.2f" % ((t1 - t0) * 1e3, m, (t2 - t1) * 1e3, (t3 - t2) * 1e3, [round(x, 2) for x in rest], (t5 - t4) * 1e3), flush=True)
shutil.rmtree(p, ignore_errors=True)
