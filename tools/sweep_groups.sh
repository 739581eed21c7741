#!/bin/bash
# lock-step batch throughput by number of batch groups: bash tools/sweep_groups.sh
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for wl in "--workload batch --batch 64" "--workload mapbuild"; do
  for g in 0 2 3 4 6 8; do
    echo -n "$wl groups=$g: "
    if [ $g = 0 ]; then python3 bench.py $wl --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"
    else NDT_BATCH_GROUPS=$g python3 bench.py $wl --no-cpu-baseline --no-mapbuild-leg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1))"; fi
  done
done
