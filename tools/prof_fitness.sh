#!/bin/bash
# kernel stats of getFitnessScore on the reference pair: bash tools/prof_fitness.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_fitness
rm -rf $O; mkdir -p $O
cat > /tmp/fit_case.py <<PY
import sys, os, numpy as np
sys.path.insert(0, "$R")
from toyslam_amd import ndt
d = np.load("$R/tests/golden/pair_0p1.npz")
t, s = d["target"], d["source"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
g.setInputTarget(t); g.setInputSource(s); g.align()
for i in range(20): g.getFitnessScore()
for i in range(5):
    g.setInputTarget(t); g.getFitnessScore()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 /tmp/fit_case.py > /dev/null 2>&1
cd $R && python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/prof_fitness/*/*kernel_stats.csv"))[-1]
for r in list(csv.DictReader(open(f)))[:16]:
    print(r["Name"][:80].ljust(80), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us")
PY
