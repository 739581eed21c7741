#!/bin/bash
# one verification pass on a GPU box: every fuzzer, the soak and leak checks, the switch matrix:
#   gpurun --timeout 1200 -- bash tools/verify_all.sh        (logs under gpurun_out/verify/)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/verify
mkdir -p $O
set -e
timeout -k 10 300 python tools/fuzz_grid.py 7 120 > $O/fuzz_grid.log 2>&1;   tail -1 $O/fuzz_grid.log
FUZZ_GRID_INDEX=2 timeout -k 10 300 python tools/fuzz_grid.py 8 60 > $O/fuzz_grid_sparse.log 2>&1; tail -1 $O/fuzz_grid_sparse.log
timeout -k 10 300 python tools/fuzz_align.py 7 40 > $O/fuzz_align.log 2>&1;  tail -1 $O/fuzz_align.log
timeout -k 10 300 python tools/fuzz_batch.py 7 30 > $O/fuzz_batch.log 2>&1;  tail -1 $O/fuzz_batch.log
timeout -k 10 300 python tools/fuzz_paths.py 7 40 > $O/fuzz_paths.log 2>&1;  tail -1 $O/fuzz_paths.log
timeout -k 10 300 python tools/fuzz_stateful.py 7 200 > $O/fuzz_stateful.log 2>&1; tail -1 $O/fuzz_stateful.log
timeout -k 10 300 python tools/fuzz_fitness.py 7 60 > $O/fuzz_fitness.log 2>&1; tail -1 $O/fuzz_fitness.log
timeout -k 10 300 python tools/fuzz_gicp.py 40 7 > $O/fuzz_gicp.log 2>&1;    tail -1 $O/fuzz_gicp.log
timeout -k 10 300 python tools/soak.py 60 > $O/soak.log 2>&1;              tail -1 $O/soak.log
timeout -k 10 300 python tools/leak_check.py > $O/leak.log 2>&1;           tail -1 $O/leak.log
bash tools/test_switches.sh > $O/switches.log 2>&1; cat $O/switches.log
