#!/bin/bash
# A/B of two builds of the library on one GPU box: bash tools/ab_lib.sh <other.so> [bench args...]
# runs the bench alternately with toyslam_amd/libndt_mi355.so (A) and the other build (B), three times each
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
other=$1; shift
cd $R
cp toyslam_amd/libndt_mi355.so /tmp/lib_A.so
cp $other /tmp/lib_B.so
for i in 1 2 3; do
  for v in A B; do
    cp /tmp/lib_$v.so toyslam_amd/libndt_mi355.so
    python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4))"
  done
done
cp /tmp/lib_A.so toyslam_amd/libndt_mi355.so
