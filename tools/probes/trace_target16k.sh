#!/bin/bash
# kernel trace of repeated setInputTarget(host cloud, 16 k points): gaps between the kernels of consecutive builds
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/trace16k; rm -rf $O; mkdir -p $O
cat > /tmp/t16.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$R")
from toyslam_amd import ndt
d = np.load(os.path.join("$R", "tests", "golden", "pair_0p1.npz"))
t = d["target"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
for i in range(40): g.setInputTarget(t)
import torch; torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace -d $O --output-format csv -- python3 /tmp/t16.py > /dev/null 2>&1
cd $R && python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/trace16k/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-24:]
t0 = int(rows[0]["Start_Timestamp"])
prev_end = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print("%-28s start %8.1f us  dur %6.1f us  gap %6.1f us" % (r["Kernel_Name"].split("(")[0][-28:], (s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end else 0.0))
    prev_end = e
PY
