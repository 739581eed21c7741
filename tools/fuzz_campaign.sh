#!/bin/bash
# a longer randomised campaign with seeds of its own (the fixed seeds of verify_all.sh find what they found long ago):
#   gpurun --timeout 1200 -- bash tools/fuzz_campaign.sh <first seed> [seeds]      (logs under gpurun_out/campaign/)
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=gpurun_out/campaign
mkdir -p $O
s0=${1:-101}; ns=${2:-3}
rc=0
for ((s = s0; s < s0 + ns; s++)); do
  timeout -k 10 280 python tools/fuzz_grid.py $s 80 > $O/grid_$s.log 2>&1 || rc=1;       echo "seed $s grid:     $(tail -1 $O/grid_$s.log)"
  FUZZ_GRID_INDEX=2 timeout -k 10 200 python tools/fuzz_grid.py $s 30 > $O/grid_sparse_$s.log 2>&1 || rc=1; echo "seed $s sparse:   $(tail -1 $O/grid_sparse_$s.log)"
  NDT_K1_SMALL_LIST=8 NDT_VF_FROM=0 timeout -k 10 200 python tools/fuzz_grid.py $s 40 > $O/grid_forced_$s.log 2>&1 || rc=1; echo "seed $s forced:   $(tail -1 $O/grid_forced_$s.log)"
  timeout -k 10 280 python tools/fuzz_align.py $s 30 > $O/align_$s.log 2>&1 || rc=1;      echo "seed $s align:    $(tail -1 $O/align_$s.log)"
  timeout -k 10 280 python tools/fuzz_batch.py $s 20 > $O/batch_$s.log 2>&1 || rc=1;      echo "seed $s batch:    $(tail -1 $O/batch_$s.log)"
  timeout -k 10 280 python tools/fuzz_paths.py $s 30 > $O/paths_$s.log 2>&1 || rc=1;      echo "seed $s paths:    $(tail -1 $O/paths_$s.log)"
  timeout -k 10 280 python tools/fuzz_stateful.py $s 150 > $O/stateful_$s.log 2>&1 || rc=1; echo "seed $s stateful: $(tail -1 $O/stateful_$s.log)"
  timeout -k 10 280 python tools/fuzz_fitness.py $s 40 > $O/fitness_$s.log 2>&1 || rc=1;  echo "seed $s fitness:  $(tail -1 $O/fitness_$s.log)"
done
exit $rc
