#!/bin/bash
# A/B of two builds of the library on the per-scan sequence (tools/time_pair.py): bash tools/ab_lib_pair.sh <other.so>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp toyslam_amd/libndt_mi355.so /tmp/lib_A.so
cp $1 /tmp/lib_B.so
for i in 1 2 3; do
  for v in A B; do
    cp /tmp/lib_$v.so toyslam_amd/libndt_mi355.so
    echo $v; python3 tools/time_pair.py 2>/dev/null | head -4
  done
done
cp /tmp/lib_A.so toyslam_amd/libndt_mi355.so
