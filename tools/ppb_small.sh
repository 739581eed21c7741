#!/bin/bash
# points per block of the latency kernels on small scans (GPU box): the reference pair (16 k points) and scans of 16 k - 64 k points
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2; do
  for v in 0 64 128 192; do
    echo "NDT_K2_PPB=$v pair: $(NDT_K2_PPB=$v python3 tools/time_pair.py 2>/dev/null | grep '^align')"
  done
done
for i in 1 2; do
  for v in 0 64 128 192; do
    NDT_K2_PPB=$v python3 tools/time_align_sizes.py 16000 30000 48000 2>/dev/null | tail -n 1
  done
done
