"""How fast the PCD reader parses 2 M-point binary files on this host: one thread and several, into pageable and into
page-locked memory (development aid):  python3 tools/probes/pcd_read_rate.py [threads]"""
import sys, time, os, numpy as np, ctypes as C, threading, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from toyslam_amd import clouds
from toyslam_amd._lib import lib
L = lib()
n = 2000000
p = tempfile.mkdtemp(prefix="pcdrate_")
rng = np.random.default_rng(1)
for k in range(4):
    clouds.write_pcd_xyz("%s/cloud_%d.pcd" % (p, k + 1), rng.standard_normal((n, 3)).astype(np.float32))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 5
pinned = None
try:
    import torch
    if torch.cuda.is_available():
        pinned = [torch.empty((n, 4), dtype=torch.float32).pin_memory() for _ in range(T)]
except Exception:
    pass
def work(tid, reps, buf_addr):
    nn = C.c_size_t(0); dense = C.c_int(0)
    for r in range(reps):
        f = "%s/cloud_%d.pcd" % (p, (r + tid) % 4 + 1)
        L.ndt_pcd_read_xyz(os.fsencode(f), buf_addr, n, 16, C.byref(nn), C.byref(dense))
for kind in ("pageable", "page-locked"):
    if kind == "page-locked" and pinned is None:
        continue
    bufs = [np.zeros((n, 4), np.float32) for _ in range(T)] if kind == "pageable" else pinned
    addr = [b.ctypes.data if kind == "pageable" else b.data_ptr() for b in bufs]
    for threads in (1, T):
        best = 1e9
        for rnd in range(3):
            th = [threading.Thread(target=work, args=(t, 6, addr[t])) for t in range(threads)]
            t0 = time.perf_counter()
            for t in th: t.start()
            for t in th: t.join()
            best = min(best, (time.perf_counter() - t0) / (6 * threads) * 1e3)
        print("%s, %d thread(s): %.2f ms per file aggregate (mmap %s)" % (kind, threads, best, os.environ.get("NDT_PCD_MMAP", "1")), flush=True)
import shutil; shutil.rmtree(p, ignore_errors=True)
