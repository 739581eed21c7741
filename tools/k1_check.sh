#!/bin/bash
# K1 after a change (GPU box): grid parity, grid fuzz, kernel stats + HBM traffic on the uniform and the surface scene,
# phase clocks, the 10 M-point build:  gpurun --timeout 900 -- bash tools/k1_check.sh [tag]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/k1_${1:-try}
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or voxel or sparse or map or fitness or small_host or reference" > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
timeout -k 10 300 python tools/fuzz_grid.py 11 60 > $O/fuzz.log 2>&1; tail -n 2 $O/fuzz.log
bash tools/prof_k1.sh 1e6 0 1.0 new pmc > $O/prof_u.log 2>&1; cat $O/prof_u.log
bash tools/prof_k1.sh 1e6 100 1.0 new pmc > $O/prof_s.log 2>&1; cat $O/prof_s.log
bash tools/prof_k1.sh 1e7 400 0.5 new > $O/prof_10m.log 2>&1; cat $O/prof_10m.log
NDT_K1_LDS_CAP=512 timeout -k 10 300 python tools/fuzz_grid.py 12 40 > $O/fuzz_cap512.log 2>&1; tail -n 1 $O/fuzz_cap512.log
timeout -k 10 120 python tools/k1_stamps.py 1e6 0 1.0 > $O/stamps_u.log 2>&1; tail -n 30 $O/stamps_u.log
timeout -k 10 200 python tools/time_k1_forms.py > $O/forms.log 2>&1; tail -n 20 $O/forms.log
