#!/bin/bash
# VALU / wave counters of the evaluation kernels for the default bench command (separate --pmc passes; never combined
# with trace domains other than --kernel-trace).  Usage on the GPU box: bash tools/pmc_valu.sh r02
tag=${1:-r02}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_valu
rm -rf $O
mkdir -p $O
B="--steps 5 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc"
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/a --output-format csv -- python3 $R/bench.py $B > $O/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d $O/b --output-format csv -- python3 $R/bench.py $B > $O/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAIT_INST_ANY -d $O/c --output-format csv -- python3 $R/bench.py $B > $O/c.log 2>&1
cd $R && python3 - "$tag" <<'PY'
import csv, glob, collections, json, sys
tag = sys.argv[1]
out = {"command": "python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc (three separate --pmc passes)", "kernels": {}}
for name in ("a", "b", "c"):
    for f in glob.glob("gpurun_out/pmc_valu/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:90]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "k_eval_server" in k or "k_derivatives_fused" in k:
                out["kernels"].setdefault(k, {}).update({c: sum(x) / len(x) for c, x in v.items()})
# evaluations the persistent kernel served per launch in that run: from the bench line of pass a
try:
    line = [ln for ln in open("gpurun_out/pmc_valu/a.log") if ln.startswith("{")][-1]
    d = json.loads(line)
    out["evaluations_per_launch"] = d["evaluations_per_registration"] + d["f64_hessian_recomputes"]
except Exception as e:
    out["evaluations_per_launch"] = None
for k, v in out["kernels"].items():
    if "k_eval_server<7>" in k and out["evaluations_per_launch"]:
        out["server_valu_insts_per_wave_per_evaluation"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"] / out["evaluations_per_launch"]
    if "k_derivatives_fused<7, true" in k:
        out["fused_valu_insts_per_wave"] = v["SQ_INSTS_VALU"] / v["SQ_WAVES"]
json.dump(out, open("gpurun_out/%s_pmc_valu.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
