#!/bin/bash
# A/B of two builds of the library on ONE GPU box (box-to-box spread is larger than most changes):
#   here:  build the old state, cp toyslam_amd/libndt_mi355.so toyslam_amd/libndt_A.so; build the new one, cp ... libndt_B.so
#   box:   bash tools/ab_build.sh [bench args...]      (three rounds, A B A B A B; leaves B in place)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
for i in 1 2 3; do
  for v in A B; do
    cp toyslam_amd/libndt_$v.so toyslam_amd/libndt_mi355.so
    python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4), d.get('protocol_us_per_evaluation'), d.get('body_us_per_evaluation'))"
  done
done
