cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_large; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch --output-format csv -- python3 $R/bench.py --workload large --steps 2 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc --output-format csv -- python3 $R/bench.py --workload large --steps 2 --warmup 1 --no-cpu-baseline --no-mapbuild-leg --no-pmc > $O/tcc.log 2>&1
cd $R && python3 - <<'PY'
import csv, glob, collections
for name in ("fetch","tcc"):
    for f in glob.glob("gpurun_out/pmc_large/%s/*/*counter_collection.csv" % name):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            for c, x in v.items():
                print(name, k, c, "mean %.1f n %d" % (sum(x)/len(x), len(x)))
PY
rm -rf $O/fetch $O/tcc
