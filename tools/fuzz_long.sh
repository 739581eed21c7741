#!/bin/bash
# a longer fuzz pass with fresh seeds: gpurun --timeout 1200 -- bash tools/fuzz_long.sh <seed>
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
S=${1:-101}
O=gpurun_out/fuzz_long
mkdir -p $O
timeout -k 10 400 python tools/fuzz_grid.py $S 400 > $O/grid.log 2>&1; tail -1 $O/grid.log
FUZZ_GRID_INDEX=2 timeout -k 10 300 python tools/fuzz_grid.py $((S+1)) 150 > $O/grid_sparse.log 2>&1; tail -1 $O/grid_sparse.log
timeout -k 10 400 python tools/fuzz_align.py $S 120 > $O/align.log 2>&1; tail -1 $O/align.log
timeout -k 10 400 python tools/fuzz_batch.py $S 60 > $O/batch.log 2>&1; tail -1 $O/batch.log
timeout -k 10 300 python tools/fuzz_paths.py $S 120 > $O/paths.log 2>&1; tail -1 $O/paths.log
timeout -k 10 300 python tools/fuzz_stateful.py $S 500 > $O/stateful.log 2>&1; tail -1 $O/stateful.log
timeout -k 10 300 python tools/fuzz_fitness.py $S 300 > $O/fitness.log 2>&1; tail -1 $O/fitness.log
timeout -k 10 300 python tools/fuzz_gicp.py 120 $S > $O/gicp.log 2>&1; tail -1 $O/gicp.log
