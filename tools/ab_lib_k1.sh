#!/bin/bash
# A/B of two builds of the library on the 1 M-point target builds (tools/time_k1_1m.py): bash tools/ab_lib_k1.sh <other.so>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
cp toyslam_amd/libndt_mi355.so /tmp/lib_A.so
cp $1 /tmp/lib_B.so
for i in 1 2 3; do
  for v in A B; do
    cp /tmp/lib_$v.so toyslam_amd/libndt_mi355.so
    echo -n "$v "; python3 tools/time_k1_1m.py 2>/dev/null | tail -1
  done
done
cp /tmp/lib_A.so toyslam_amd/libndt_mi355.so
