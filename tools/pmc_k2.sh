#!/bin/bash
# PMC passes over the bench workload (development aid).  Separate passes per counter group.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # name counters...
  local name=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" -d $R/gpurun_out/pmc_$name --output-format csv -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$name.log 2>&1
}
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
run sq2 SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_LDS_BANK_CONFLICT
run fetch FETCH_SIZE
run write WRITE_SIZE
run tcc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum
cd $R && python3 - <<'PY'
import csv,glob,collections
for d in sorted(glob.glob('gpurun_out/pmc_*/')):
    for f in glob.glob(d+'**/*counter_collection.csv', recursive=True):
        agg=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:70]
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
        print('==',d)
        for k,v in agg.items():
            if 'k_derivatives' in k or 'k_reduce' in k or 'hessian64' in k:
                print(k, {c:(round(sum(x)/len(x),1),len(x)) for c,x in v.items()})
PY
