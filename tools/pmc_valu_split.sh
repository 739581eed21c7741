#!/bin/bash
# Where the evaluation server's vector instructions go: SQ_INSTS_VALU / SQ_INSTS_SALU / SQ_WAVES of one
# server launch that serves 305 rounds of ONE kind -- no-op rounds (command, tables, fold, fan-in, publication, polling: no
# per-point body), rounds without and with the Hessian -- at the headline workload.  One --pmc pass per kind (the profiler
# around a fresh process each time; --kernel-trace only).   bash tools/pmc_valu_split.sh [tag]
tag=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_split
rm -rf $O; mkdir -p $O
cat > /tmp/split_case.py <<PY
import sys, numpy as np
sys.path.insert(0, "$R")
from toyslam_amd import clouds, ndt
tgt = clouds.target_uniform(1000000)
src = clouds.source_from_target(tgt, 100000, seed=clouds.SEED + 1)
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0); g.setMaximumIterations(28); g.setTransformationEpsilon(1e-9)
g.setInputTarget(tgt); g.setInputSource(src); g.align()
rt = g.diag_server_roundtrip(ndt.host_matrix_to_pose(g.getFinalTransformation()), 300)
print(rt)
PY
for kind in 3 1 0; do
  NDT_DIAG_ONLY_KIND=$kind rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d $O/k$kind --output-format csv -- python3 /tmp/split_case.py > $O/k$kind.log 2>&1
done
cd $R && python3 - "$tag" <<'PY'
import csv, glob, json, sys
tag = sys.argv[1]
out = {"what": "one k_eval_server launch serving 305 rounds of one kind (ndt_diag_server_roundtrip, NDT_DIAG_ONLY_KIND), headline workload "
               "(100 k-point source, 1 M-point target, 256 blocks x 8 waves)", "rounds_per_launch": 305, "kinds": {}}
names = {3: "no_op_round", 1: "without_hessian", 0: "with_hessian"}
for kind in (3, 1, 0):
    best = None
    for f in glob.glob("gpurun_out/pmc_split/k%d/*/*counter_collection.csv" % kind):
        rows = {}
        for r in csv.DictReader(open(f)):
            if "k_eval_server" in r["Kernel_Name"]:
                rows.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
        for d in rows.values():  # the diag launch is the one with by far the most instructions
            if best is None or d.get("SQ_INSTS_VALU", 0) > best.get("SQ_INSTS_VALU", 0):
                best = d
    if best:
        w = best["SQ_WAVES"]
        out["kinds"][names[kind]] = {"SQ_INSTS_VALU_per_round": best["SQ_INSTS_VALU"] / 305, "SQ_INSTS_SALU_per_round": best["SQ_INSTS_SALU"] / 305,
                                     "waves": w, "valu_per_wave_per_round": best["SQ_INSTS_VALU"] / 305 / w}
    try:
        out["kinds"][names[kind]]["round_us_under_the_profiler"] = [ln for ln in open("gpurun_out/pmc_split/k%d.log" % kind) if ln.startswith("{")][-1].strip()
    except Exception:
        pass
k = out["kinds"]
if all(n in k for n in names.values()):
    out["body_valu_per_wave_with_hessian"] = k["with_hessian"]["valu_per_wave_per_round"] - k["no_op_round"]["valu_per_wave_per_round"]
    out["body_valu_per_wave_without_hessian"] = k["without_hessian"]["valu_per_wave_per_round"] - k["no_op_round"]["valu_per_wave_per_round"]
    out["protocol_valu_per_wave_per_round"] = k["no_op_round"]["valu_per_wave_per_round"]
json.dump(out, open("gpurun_out/%s_pmc_valu_split.json" % tag, "w"), indent=1)
print(json.dumps(out, indent=1))
PY
