#!/bin/bash
# A/B of the K2 launch shape (separate processes; development aid)
set -u
out=gpurun_out/tune_k2.log
: > $out
for split in 1 0; do
  for blocks in 256 512 1024 2048 4096; do
    for spin in 1; do
      echo "=== SPLIT=$split MAX_BLOCKS=$blocks SPIN=$spin" >> $out
      NDT_K2_VARIANT=$split NDT_K2_MAX_BLOCKS=$blocks NDT_SPIN_WAIT=$spin timeout -k 5 120 python tools/gpu_probe.py ${1:-U} 2>&1 | grep -E "eval\(|event-timed|align median" >> $out
    done
  done
done
echo "=== SPLIT=1 MAX_BLOCKS=1024 SPIN=0" >> $out
NDT_SPIN_WAIT=0 timeout -k 5 120 python tools/gpu_probe.py ${1:-U} 2>&1 | grep -E "eval\(|event-timed|align median" >> $out
cat $out
