#!/bin/bash
# A/B of one environment setting on one GPU box: bash tools/ab_env.sh VAR=value [bench args...]   (A: unset, B: set; three rounds)
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
setting=$1; shift
cd $R
for i in 1 2 3; do
  python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('A (default)', round(d['value'],1), round(d['ms_per_step'],4))"
  env $setting python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('B ($setting)', round(d['value'],1), round(d['ms_per_step'],4))"
done
