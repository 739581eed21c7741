#!/bin/bash
# GICP row: rocprofv3 kernel stats + timings of tools/time_gicp.py on the reference pair and on the
# 1M / 100k synthetic scene.  Usage (on the GPU box): tools/profile_gicp.sh r01
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${tag}_gicp
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/pair -- python3 $R/tools/time_gicp.py pair > $O/pair.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/large -- python3 $R/tools/time_gicp.py 1000000 100000 > $O/large.log 2>&1
cd $R
python3 tools/time_gicp.py pair --oracle 2>/dev/null | tail -1 > $O/time_pair.json
python3 tools/time_gicp.py 1000000 100000 --oracle 2>/dev/null | tail -1 > $O/time_large.json
python3 - "$O" <<'PY'
import csv, glob, json, sys
O = sys.argv[1]
out = {"command": "python tools/time_gicp.py {pair | 1000000 100000} [--oracle]", "timings": {}, "kernels": {}}
for name in ("pair", "large"):
    out["timings"][name] = json.load(open("%s/time_%s.json" % (O, name)))
    f = glob.glob("%s/%s/*/*kernel_stats.csv" % (O, name))[0]
    open("%s/kernel_stats_%s.csv" % (O, name), "w").write(open(f).read())
    out["kernels"][name] = {r["Name"][:90]: {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "max_us": float(r["MaxNs"]) / 1e3,
                                            "pct": float(r["Percentage"])} for r in list(csv.DictReader(open(f)))[:8]}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
PY
