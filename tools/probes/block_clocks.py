"""configs[2] evaluation at the identity pose: when each block of k_derivatives_fused started and when its first / last wave
left the per-point body (library built with -DNDT_DIAG_BLOCK_CLOCKS; s_memtime, 100 MHz) -- how even is the work?"""
import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from toyslam_amd import clouds, ndt
from toyslam_amd._lib import lib
L = lib()
tgt = clouds.target_surfaces(10000000, extent=400.0, n_boxes=60)
src = clouds.source_from_target(tgt, 2000000, seed=clouds.SEED + 1)
g = ndt.NormalDistributionsTransform(); g.setResolution(0.5)
g.setInputTarget(tgt); g.setInputSource(src)
p = np.zeros(6)
for _ in range(5): g.eval(p)
nb = 512
out = np.zeros((nb, 4), dtype=np.uint64)
L.ndt_diag_block_clocks_read.argtypes = [C.c_void_p, C.c_int]
rc = L.ndt_diag_block_clocks_read(out.ctypes.data, nb)
t0 = out[:, 0].astype(np.int64); lo = out[:, 1].astype(np.int64); hi = out[:, 2].astype(np.int64); xcc = out[:, 3].astype(np.int64) & 15
np.save(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "block_clocks.npy"), out)
b = t0.min()
u = lambda a: (a - b) / 100.0  # us
print("kernel: first block start -> last wave end %.1f us" % u(hi).max())
print("block starts us: min %.1f med %.1f p90 %.1f max %.1f" % tuple(np.percentile(u(t0), [0, 50, 90, 100])))
print("last-wave end us: p0 %.1f p10 %.1f p25 %.1f med %.1f p75 %.1f p90 %.1f max %.1f" % tuple(np.percentile(u(hi), [0, 10, 25, 50, 75, 90, 100])))
print("first-wave end us: p0 %.1f med %.1f max %.1f" % tuple(np.percentile(u(lo), [0, 50, 100])))
print("mean block busy %.1f us = %.0f %% of the span" % ((hi - t0).mean() / 100.0, 100 * (hi - t0).mean() / (hi.max() - b)))
for x in range(8):
    m = (np.arange(nb) % 8) == x
    print("XCD %d: last-wave end med %.1f max %.1f us; block busy mean %.1f" % (x, np.median(u(hi)[m]), u(hi)[m].max(), (hi - t0)[m].mean() / 100.0))
