#!/bin/bash
# HBM / L2 counters of the configs[2] evaluation kernel (separate --pmc passes).  Usage on the GPU box: bash tools/pmc_large.sh [extent]
ext=${1:-400}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_large
rm -rf $O
mkdir -p $O
B="--workload large --extent $ext --steps 3 --warmup 1 --no-cpu-baseline --no-mapbuild-leg"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py $B > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py $B > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- python3 $R/bench.py $B > $O/tcc.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/req --output-format csv -- python3 $R/bench.py $B > $O/req.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name in ("fetch", "write", "tcc", "req"):
    for f in glob.glob("gpurun_out/pmc_large/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "k_eval_server" in k or "k_derivatives_fused" in k:
                out.setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
line = [ln for ln in open("gpurun_out/pmc_large/fetch.log") if ln.startswith("{")][-1]
d = json.loads(line)
out["evaluations_per_launch"] = d["evaluations_per_registration"] + d["f64_hessian_recomputes"]
out["mean_neighbors"] = d["mean_neighbors"]
print(json.dumps(out, indent=1))
PY
