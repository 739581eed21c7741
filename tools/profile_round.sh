#!/bin/bash
# Round profile: rocprofv3 kernel stats + PMC (FETCH_SIZE / WRITE_SIZE in separate passes) of the
# default bench command.  Usage (on the GPU box): tools/profile_round.sh r01
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
rm -rf $O
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/stats --output-format csv -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/write.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/tcc.log 2>&1
cd $R && python3 - "$tag" <<'PY'
import csv, glob, json, sys, collections
tag = sys.argv[1]
O = "gpurun_out/%s" % tag
out = {"command": "python bench.py --steps K --warmup W --no-cpu-baseline --no-mapbuild-leg --no-pmc (stats: K=20; each --pmc pass on its own: K=5)", "kernels": {}, "pmc": {}}
f = glob.glob(O + "/stats/*/*kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
open(O + "/kernel_stats.csv", "w").write(open(f).read())
for r in rows[:28]:
    calls = int(r["Calls"])
    out["kernels"][r["Name"][:100]] = {"calls": calls, "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"]),
                                       "min_us": float(r["MinNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                       # the process's first launch of a kernel is cold (code object load, first touch of the
                                       # mailbox pages): the mean without the one slowest call is what bench.py's timed region sees
                                       "avg_without_slowest_us": (float(r["TotalDurationNs"]) - float(r["MaxNs"])) / 1e3 / max(calls - 1, 1)}
for name in ("fetch", "write", "tcc"):
    for f in glob.glob(O + "/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if any(t in k for t in ("k_derivatives", "k_eval_server", "k_hessian64", "k1_", "k_bbox", "k_repack", "k_rc_")):
                out["pmc"].setdefault(k, {}).update({c: {"mean": sum(x) / len(x), "n": len(x)} for c, x in v.items()})
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
PY
