#!/bin/bash
out=gpurun_out/diag_k2.log
: > $out
for mode in 0 1 2 3; do
  echo "=== DBG_MODE=$mode (non-split)" >> $out
  NDT_K2_SPLIT=0 NDT_DBG_MODE=$mode timeout -k 5 120 python tools/gpu_probe.py ${1:-U} 2>&1 | grep -E "eval\(|event-timed" >> $out
done
cat $out
