#!/bin/bash
# the rows around the path after a change (GPU box): cloud API + N1 / N2 parity, the node loop app, its profile
#   gpurun --timeout 900 -- bash tools/node_check.sh [tag]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/node_${1:-try}
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_adapter.py -x -q -m gpu -k "resident or voxel_grid_filter or map_ or map_sequence or small_host or grid" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -n 5 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 300 python tools/node_loop_profile.py 40 60000 node $O/node_loop.json > $O/node_loop.log 2>&1; python - <<PY
import json
r = json.load(open("$O/node_loop.json"))
for k in ("device_resident", "host_clouds"):
    print(k, r.get(k, {}).get("ms_per_scan"), r.get(k, {}).get("line"))
print("kernel us per scan", r.get("kernel_us_per_scan"), "launches per scan", r.get("launches_per_scan"))
for x in (r.get("kernels") or [])[:16]:
    print("  %-70s calls %5d avg %7.2f us  per scan %7.2f us" % (x["name"][:70], x["calls"], x["avg_us"], x["total_us_per_scan"]))
PY
