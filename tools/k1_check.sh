#!/bin/bash
# K1 after a change (GPU box): the grid tests, the randomised grid check, build times per cloud shape, kernel stats at 1 M points
#   gpurun --timeout 900 -- bash tools/k1_check.sh
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/k1
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or voxel or sparse or map or fitness or small_host or reference" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 300 python tools/fuzz_grid.py 11 60 > $O/fuzz.log 2>&1; tail -n 2 $O/fuzz.log
bash tools/prof_k1.sh 1e6 0 1.0 > $O/prof_u.log 2>&1; cat $O/prof_u.log
