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
# roofline (HBM, 8 TB/s) of the three GICP kernels from their longest call in each run: algorithmic bytes as DESIGN.md section 9
# states them -- covariances n (16 + 16 k + 48), correspondence step n (16 + 16 + 2 * 48 + 40), objective m (16 + 16 + 4 + 36)
out["roofline"] = {}
for name in ("pair", "large"):
    t = out["timings"][name]
    nt, ns, k, m = t["n_target"], t["n_source"], 20, t["n_source"]
    rows = {r: v for r, v in out["kernels"][name].items()}
    def pick(prefix):
        for r, v in rows.items():
            if prefix in r:
                return v
        return None
    rl = {}
    # the objective server serves one BFGS run per launch: evaluations / outer iterations of them, m * 72 B each
    per_launch = max(1.0, t["evaluations"] / max(1, t["iterations"]))
    for kern, nbytes, which in (("k_knn_covariances", nt * (16 + 16 * k + 48), "max_us"), ("k_correspond", ns * (16 + 16 + 96 + 40), "max_us"),
                                ("k_gicp_server", m * 72 * per_launch, "avg_us"), ("k_functor<2>", m * 72, "avg_us")):
        v = pick(kern)
        if v:
            gbs = nbytes / (v[which] * 1e-6) / 1e9
            rl[kern] = {"bound": "hbm", "algorithmic_bytes": nbytes, "kernel_us": v[which], "achieved": gbs, "peak": 8000.0, "unit": "GB/s", "frac": gbs / 8000.0}
    out["roofline"][name] = rl
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
PY
