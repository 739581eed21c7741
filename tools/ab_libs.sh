#!/bin/bash
# A/B of several builds of the library on ONE GPU box: bash tools/ab_libs.sh A max-ilp ... -- [bench args]   (toyslam_amd/libndt_<name>.so; leaves the first in place)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" = "--" ] && shift
for i in 1 2 3; do
  for v in "${names[@]}"; do
    cp toyslam_amd/libndt_$v.so toyslam_amd/libndt_mi355.so
    python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['value'],1), round(d['ms_per_step'],4), d.get('protocol_us_per_evaluation'), d.get('body_us_per_evaluation'))"
  done
done
cp toyslam_amd/libndt_${names[0]}.so toyslam_amd/libndt_mi355.so
