#!/bin/bash
# A/B of one environment setting on the mapping nodes' per-scan sequence (tools/time_pair.py): bash tools/ab_pair.sh VAR=value
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for i in 1 2 3; do
  echo "A (default)"; python3 tools/time_pair.py 2>/dev/null | head -4
  echo "B ($1)"; env $1 python3 tools/time_pair.py 2>/dev/null | head -4
done
