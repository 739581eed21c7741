#!/bin/bash
for i in 1 2 3; do
  NDT_BENCH_THREADS=1 NDT_TIMING=2 python bench.py --workload large --extent 200 --steps 10 --warmup 2 --no-cpu-baseline --no-mapbuild-leg > gpurun_out/l200_$i.json 2> gpurun_out/l200_$i.err
  grep -n -E "server left|bench threads" gpurun_out/l200_$i.err | cut -c1-300
  python -c "import json; d=json.load(open('gpurun_out/l200_$i.json')); print(d['value'], d['ms_per_step'])"
done
grep -E "nr_thr|throttled" /sys/fs/cgroup/cpu.stat
echo "--- 64 BLAS threads"
OPENBLAS_NUM_THREADS=64 NDT_TIMING=2 python bench.py --workload large --extent 200 --steps 10 --warmup 2 --no-cpu-baseline --no-mapbuild-leg 2>&1 >/dev/null | grep -E "server left|gap=[0-9]{5}"
grep -E "nr_thr|throttled" /sys/fs/cgroup/cpu.stat
