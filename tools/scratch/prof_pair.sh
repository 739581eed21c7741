#!/bin/bash
# kernel stats of tools/time_pair.py under a K1 form: bash tools/scratch/prof_pair.sh new|old
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/k1pair_$1
cd /tmp && export TMPDIR=/tmp
NDT_K1=$1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/k1pair_$1 --output-format csv -- python3 $R/tools/time_pair.py > /dev/null 2>&1
cd $R && python3 - "$1" <<'PY'
import csv,glob,sys
f=sorted(glob.glob("gpurun_out/k1pair_%s/*/*kernel_stats.csv" % sys.argv[1]))[-1]
for r in list(csv.DictReader(open(f)))[:18]:
    print(r["Name"][:60].ljust(60), r["Calls"], r["AverageNs"], r["MinNs"], r["MaxNs"])
PY
