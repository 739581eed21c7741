#!/bin/bash
# SQ counters of the big-scan derivative kernel (--workload large):  bash tools/pmc_large_sq.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_large_sq; rm -rf $O; mkdir -p $O
B="--workload large --steps 2 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc"
run() { rocprofv3 --kernel-trace --pmc $2 -d $O/$1 --output-format csv -- python3 $R/bench.py $B > $O/$1.log 2>&1; }
run a "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_RD" &&
run b "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" &&
run c "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" &&
run d "SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_SALU SQ_WAIT_ANY GRBM_GUI_ACTIVE" &&
run e "SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_ACTIVE_INST_MISC"
cd $R && python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_large_sq/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_derivatives_fused" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][30:75]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {cn: round(sum(x) / len(x)) for cn, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
for p in a b c d e; do rm -rf $O/$p; done
