#!/bin/bash
# rocprofv3 kernel stats + HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes) of the lock-step batch workload:
#   gpurun -- bash tools/profile_batch.sh      -> gpurun_out/prof_batch/{stats,fetch,write}
set -e
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$R/gpurun_out/prof_batch
rm -rf $O
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --workload batch --batch 64 --steps 3 --warmup 1 > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --workload batch --batch 64 --steps 1 --warmup 1 > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --workload batch --batch 64 --steps 1 --warmup 1 > $O/write.log 2>&1
find $O -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/kernel_stats.csv
ls $O/*
