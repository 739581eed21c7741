#!/bin/bash
# Memory-side counters of the big-scan derivative kernel (--workload large): L2 hits, L1 -> L2 read latency, address translation,
# the texture addresser's stalls.  One rocprofv3 pass per counter group:  bash tools/pmc_large_mem.sh [extra bench flags]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_large_mem; rm -rf $O; mkdir -p $O
B="--workload large --steps 2 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc $*"
run() { rocprofv3 --kernel-trace --pmc $2 -d $O/$1 --output-format csv -- python3 $R/bench.py $B > $O/$1.log 2>&1; }
run p1 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" &&
run p2 "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCP_LATENCY_sum" &&
run p3 "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_PENDING_STALL_CYCLES_sum" &&
run p4 "TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" &&
run p5 "SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_ANY SQ_WAVE_CYCLES" &&
run p6 "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum TCC_BUSY_avr GRBM_GUI_ACTIVE"
rc=$?
cd $R && python3 - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_large_mem/p*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_derivatives_fused" in r["Kernel_Name"] or "k_hessian64" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][30:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print(k, {cn: round(sum(x) / len(x)) for cn, x in v.items()}, "launches", len(next(iter(v.values()))))
PY
for p in p1 p2 p3 p4 p5 p6; do rm -rf $O/$p; done
exit $rc
