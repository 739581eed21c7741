"""What would a device-side Newton / More-Thuente driver start from?  Per-evaluation time of the evaluation server as
it is (host posts every command, ndt_diag_server_roundtrip) against the same round driven from the device
(ndt_diag_selfdrive: the last arriving block adds the part sums and posts the next command itself; no solver step).
  python tools/selfdrive_probe.py [source_points]"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from toyslam_amd import clouds, ndt

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 100000
tgt = clouds.target_uniform(1000000)
src = clouds.source_from_target(tgt, n)
g = ndt.NormalDistributionsTransform()
g.setInputTarget(tgt)
g.setInputSource(src)
p = np.array([0.3, -0.2, 0.1, 0.0087, -0.0052, 0.0175])
ref = g.eval(p, True)
host = g.diag_server_roundtrip(p, 300)
dev = g.diag_selfdrive(p, 300)
chk = g.eval(p, True)
out = {"source_points": n, "host_driven_us": host, "device_driven_us": dev,
       "score_of_reference_evaluation": ref[0] if isinstance(ref, tuple) else None}
print(json.dumps(out, default=float))
