#!/bin/bash
# the GPU parity suite under every development switch (README table): bash tools/test_switches.sh
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for sw in "NDT_PERSISTENT=0" "NDT_MAILBOX=host" "NDT_SERVER_DIRECT=0" "NDT_K2_FUSED=0" "NDT_SPIN_WAIT=0" "NDT_K1=old" "NDT_K1_COMPACT=eager" "NDT_K1_COMPACT=off" "NDT_K1_LDS_CAP=512" "NDT_K1_BUCKETS=768" "NDT_BATCH_GROUPS=1" "NDT_BATCH_GROUPS=3" "NDT_SORT_SOURCE=1" "NDT_SORT_SOURCE=0" "NDT_K2_PPB=512" "NDT_GICP_SERVER=0" "NDT_HOST_STAGE_MAX=0" "NDT_BBOX_POLL=0" "NDT_K1_SMALL=0" "NDT_K1_SMALL_LIST=8" "NDT_K1_SMALL_FINISH=0" "NDT_K1_INDEX=1" "NDT_ZERO_COPY=1" "NDT_VF_FROM=0" "NDT_VF=chain" "NDT_HOST_AVX2=0" "NDT_PCD_MMAP=0" "NDT_PINNED_CACHE_MB=0" "NDT_ORDER=chain" "NDT_ORDER_RADIX_FROM=0"; do
  printf "%-22s " "$sw"
  env $sw timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gicp_gpu.py -x -q -m gpu 2>&1 | tail -1
done
