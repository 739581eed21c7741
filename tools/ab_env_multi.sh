#!/bin/bash
# several settings of one environment variable against the default on ONE GPU box: bash tools/ab_env_multi.sh VAR v1 v2 ... -- [bench args]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd $R
var=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" = "--" ] && shift
show='import json,sys; d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],4), d.get("protocol_us_per_evaluation"), d.get("body_us_per_evaluation"))'
for i in 1 2 3; do
  python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "$show" "default"
  for v in "${vals[@]}"; do
    env $var=$v python3 bench.py "$@" --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "$show" "$var=$v"
  done
done
