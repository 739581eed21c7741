#!/bin/bash
# one development switch under the GPU parity suites, verbosely and bounded (which test fails / hangs): bash tools/switch_debug.sh VAR=value [seconds]
cd ${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p gpurun_out/switch
f=gpurun_out/switch/$(echo "$1" | tr '=' '_').log
env $1 timeout -k 10 ${2:-200} python -m pytest tests/test_gpu_parity.py tests/test_gicp_gpu.py -x -v -m gpu > $f 2>&1
echo "rc=$?" >> $f
grep -E "FAILED|Error|assert|rc=" $f | head -20
tail -n 5 $f
