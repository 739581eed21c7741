#!/bin/bash
# phase clocks of k1_scatter / k1_finalize on four scenes (GPU box): bash tools/k1_stamps_all.sh
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/k1_stamps
for c in "1e6 0 1.0" "1e6 100 1.0" "1e6 60 1.0" "1e7 400 0.5"; do
  echo "== $c"
  timeout -k 10 200 python tools/k1_stamps.py $c 2>&1 | grep clocks | tail -n 2
done | tee gpurun_out/k1_stamps/all.log
