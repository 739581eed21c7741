#!/bin/bash
# 1 M-point target builds by points per k1_hist / k1_scatter block: bash tools/sweep_k1_ppb.sh
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for p in 0 1024 2048 4096 8192; do
  echo -n "NDT_K1_PPB=$p "; NDT_K1_PPB=$p python3 tools/time_k1_1m.py 2>/dev/null | tail -1
done
