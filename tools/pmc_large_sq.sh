#!/bin/bash
# SQ counters of the big-scan derivative kernel (--workload large):  bash tools/pmc_large_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_large_sq; rm -rf $O; mkdir -p $O
B="--workload large --steps 2 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc"
for c in 0; do
  NDT_COMPACT=$c rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD -d $O/a$c --output-format csv -- python3 $R/bench.py $B > $O/a$c.log 2>&1
  NDT_COMPACT=$c rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU -d $O/b$c --output-format csv -- python3 $R/bench.py $B > $O/b$c.log 2>&1
done
cd $R && python3 - <<'PY'
import csv, glob, collections
for c in (0,):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for name in ("a", "b"):
        for f in glob.glob("gpurun_out/pmc_large_sq/%s%d/*/*counter_collection.csv" % (name, c)):
            for r in csv.DictReader(open(f)):
                if "k_derivatives_fused" in r["Kernel_Name"]:
                    agg[r["Kernel_Name"][30:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        print(k, {cn: round(sum(x) / len(x)) for cn, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
rm -rf $O/a0 $O/a1 $O/b0 $O/b1
