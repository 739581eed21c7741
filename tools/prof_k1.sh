#!/bin/bash
# kernel stats of target builds of one shape: bash tools/prof_k1.sh <points> <extent|0 = uniform> <resolution> [NDT_K1 mode]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_k1
rm -rf $O; mkdir -p $O
cat > /tmp/k1_case.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$R")
from toyslam_amd import clouds, ndt
import torch
n, ext, res = int(float(sys.argv[1])), float(sys.argv[2]), float(sys.argv[3])
tgt = clouds.target_uniform(n) if ext == 0 else clouds.target_surfaces(n, extent=ext, n_boxes=40)
dev = torch.from_numpy(np.c_[tgt, np.ones(n, np.float32)]).cuda()
g = ndt.NormalDistributionsTransform(); g.setResolution(res)
for i in range(6): g.setInputTargetDeviceRef(dev.data_ptr(), n)
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
NDT_K1=${4:-new} rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 /tmp/k1_case.py $1 $2 $3 > /dev/null 2>&1
cd $R && python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_k1/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:10]:
    print(r["Name"][:70].ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
