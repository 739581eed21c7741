"""60 back-to-back registrations of a 2M-point scan (configs[2] shape): median / max wall time and the registrations
that took more than 10 ms -- the probe that found the CFS-quota stalls of the polling host thread.
  python tools/stall_probe.py [extent_m]     NDT_TIMING=2 adds the longest poll gap per registration on stderr."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt
if os.environ.get('PROBE_TORCH'):
    import torch; torch.zeros(1).cuda(); torch.cuda.synchronize()
ext = float(sys.argv[1]) if len(sys.argv) > 1 else 200.0
tgt = clouds.target_surfaces(10000000, extent=ext, n_boxes=60 if ext >= 300 else 120)
src = clouds.source_from_target(tgt, 2000000)
g = ndt.NormalDistributionsTransform(); g.setResolution(0.5); g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
g.setInputTarget(tgt); g.setInputSource(src)
ts = []
if os.environ.get('PROBE_SLEEP'): time.sleep(float(os.environ['PROBE_SLEEP']))
for i in range(60):
    t0 = time.perf_counter(); g.align(); ts.append(time.perf_counter() - t0)
ts = np.array(ts) * 1e3
print("ext", ext, "env", {k: v for k, v in os.environ.items() if k.startswith("NDT_")}, "median %.2f ms  max %.2f  stalls>10ms: %d  at %s" % (np.median(ts), ts.max(), int((ts > 10).sum()), np.nonzero(ts > 10)[0].tolist()), g.stats())
