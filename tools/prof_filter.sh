#!/bin/bash
# kernel stats of the voxel filter (N1, device in / device out): bash tools/prof_filter.sh <points> <leaf> [pmc]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_filter
rm -rf $O; mkdir -p $O
cat > /tmp/vf_case.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from toyslam_amd import clouds, ndt
import torch
n, leaf = int(float(sys.argv[1])), float(sys.argv[2])
rng = np.random.default_rng(3)
world = clouds.target_surfaces(4 * n, seed=77, extent=60.0)[:, :3].astype(np.float32)
scan = (world[rng.choice(len(world), n, replace=False)] + rng.normal(0, 0.01, (n, 3))).astype(np.float32)
dev = torch.from_numpy(np.c_[scan, np.ones(n, np.float32)]).cuda()
dout = torch.empty((n, 4), dtype=torch.float32, device="cuda")
g = ndt.NormalDistributionsTransform()
for i in range(8): g.voxelGridFilterDevice(dev.data_ptr(), n, 16, leaf, dout.data_ptr())
torch.cuda.synchronize()
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O --output-format csv -- python3 /tmp/vf_case.py $1 $2 > /dev/null 2>&1
if [ "$3" = "pmc" ]; then
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 /tmp/vf_case.py $1 $2 > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 /tmp/vf_case.py $1 $2 > /dev/null 2>&1
fi
cd $R && python3 - <<'PY'
import csv, glob, collections
f = [x for x in sorted(glob.glob("gpurun_out/prof_filter/*/*kernel_stats.csv")) if "/fetch/" not in x and "/write/" not in x][-1]
mb = collections.defaultdict(lambda: [0.0, 0.0, 0])
for which, name in ((0, "fetch"), (1, "write")):
    for cf in glob.glob("gpurun_out/prof_filter/%s/*/*counter_collection.csv" % name):
        for r in csv.DictReader(open(cf)):
            e = mb[r["Kernel_Name"][:70]]
            e[which] += float(r["Counter_Value"]) * (2.0 if which == 0 else 1.0) * 1024 / 1e6
            if which == 0: e[2] += 1
tot = 0.0
for r in list(csv.DictReader(open(f)))[:14]:
    k = r["Name"][:70]
    extra = ""
    if k in mb and mb[k][2]:
        extra = "  read %.1f MB  written %.1f MB" % (mb[k][0] / mb[k][2], mb[k][1] / mb[k][2])
    per_call = float(r["TotalDurationNs"]) / 8 / 1e3
    tot += per_call
    print(k.ljust(70), r["Calls"], round(float(r["AverageNs"]) / 1e3, 1), "us  per filter %.1f" % per_call + extra)
print("kernels per filter call: %.1f us" % tot)
PY
