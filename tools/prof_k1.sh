#!/bin/bash
# kernel stats of target builds of one shape: bash tools/prof_k1.sh <points> <extent|0 = uniform> <resolution> [NDT_K1 mode] [pmc]
# (pmc: two more passes, FETCH_SIZE and WRITE_SIZE on their own -> HBM MB per kernel, 2 x FETCH_SIZE on gfx950)
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
if [ "$5" = "pmc" ]; then
  NDT_K1=${4:-new} rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 /tmp/k1_case.py $1 $2 $3 > /dev/null 2>&1
  NDT_K1=${4:-new} rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 /tmp/k1_case.py $1 $2 $3 > /dev/null 2>&1
fi
cd $R && python3 - <<'PY'
import csv, glob, collections
f = [x for x in sorted(glob.glob("gpurun_out/prof_k1/*/*kernel_stats.csv")) if "/fetch/" not in x and "/write/" not in x][-1]
mb = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, name in ((0, "fetch"), (1, "write")):
    for cf in glob.glob("gpurun_out/prof_k1/%s/*/*counter_collection.csv" % name):
        for r in csv.DictReader(open(cf)):
            e = mb[r["Kernel_Name"][:70]]
            e[which] += float(r["Counter_Value"]) * (2.0 if which == 0 else 1.0) * 1024 / 1e6
            if which == 0: e[2] += 1
total_us = 0.0
for r in list(csv.DictReader(open(f)))[:10]:
    k = r["Name"][:70]
    extra = ""
    if k in mb and mb[k][2]:
        extra = "  read %.1f MB  written %.1f MB" % (mb[k][0] / mb[k][2], mb[k][1] / mb[k][2])
    print(k.ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us" + extra)
PY
