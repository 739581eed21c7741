#!/bin/bash
# k1_small after a change (GPU box): grid parity, grid fuzz in its three regimes (lists, forced overflow, small passes), timing
# against the chain:  gpurun --timeout 900 -- bash tools/small_check.sh [tag]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/small_${1:-try}
mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py tests/test_adapter.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log; tail -n 3 $O/tests.log
grep -q "rc=0" $O/tests.log || exit 1
FUZZ_GRID_INDEX=1 timeout -k 10 300 python tools/fuzz_grid.py 21 60 > $O/fuzz.log 2>&1; tail -n 2 $O/fuzz.log
FUZZ_GRID_INDEX=1 NDT_K1_SMALL_LIST=8 timeout -k 10 300 python tools/fuzz_grid.py 22 40 > $O/fuzz_list8.log 2>&1; tail -n 1 $O/fuzz_list8.log
FUZZ_GRID_INDEX=1 NDT_K1_LDS_CAP=512 timeout -k 10 300 python tools/fuzz_grid.py 23 40 > $O/fuzz_cap512.log 2>&1; tail -n 1 $O/fuzz_cap512.log
FUZZ_GRID_INDEX=1 NDT_K1_SMALL_FINISH=0 timeout -k 10 300 python tools/fuzz_grid.py 24 40 > $O/fuzz_finish0.log 2>&1; tail -n 1 $O/fuzz_finish0.log
timeout -k 10 60 python tools/k1_stamps.py 16000 40 1.0 2>&1 | tail -n 2
K1_CASES=4 NDT_K1_SMALL=0 timeout -k 10 120 python tools/time_k1_forms.py > $O/forms_chain.log 2>&1; cat $O/forms_chain.log
K1_CASES=4 timeout -k 10 120 python tools/time_k1_forms.py > $O/forms_small.log 2>&1; cat $O/forms_small.log
timeout -k 10 120 python tools/time_pair.py > $O/pair.log 2>&1; tail -n 5 $O/pair.log
