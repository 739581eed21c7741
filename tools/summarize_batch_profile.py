#!/usr/bin/env python3
"""gpurun_out/prof_batch (tools/profile_batch.sh: rocprofv3 --stats and separate --pmc FETCH_SIZE / WRITE_SIZE passes of
`bench.py --workload batch --batch 64`) -> profiles/<tag>_batch64_summary.json + <tag>_batch64_kernel_stats.csv.

HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE and WRITE_SIZE count KB; gfx950 reports half the fetched bytes
(2 x FETCH_SIZE)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
O = os.path.join(R, "gpurun_out", "prof_batch")
out = {"command": "python bench.py --workload batch --batch 64 (stats: --steps 3 --warmup 1; each --pmc pass on its own: --steps 1 --warmup 1)",
       "note": "the timed batch runs as independent lock-step groups (2 at 64 scans); bench.py's roofline leg runs one more pass as ONE "
               "lock-step loop with an event pair around the derivative kernels of every lock-step", "kernel_stats": {}, "pmc": {}}
rows = list(csv.DictReader(open(os.path.join(O, "kernel_stats.csv"))))
for r in rows[:16]:
    out["kernel_stats"][r["Name"][:100]] = {"calls": int(r["Calls"]), "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])}
tot = collections.defaultdict(float)
for name in ("fetch", "write"):  # (one run per pass: profile_batch.sh starts from an empty directory)
    # (gpurun merges every call's files into the local gpurun_out/: only the newest run of each pass counts)
    for f in sorted(glob.glob(os.path.join(O, name, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:100]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            if "k_derivatives" in k or "k_batch_step" in k or "k_hessian64" in k:
                for c, x in v.items():
                    out["pmc"].setdefault(k, {})[c] = {"mean": sum(x) / len(x), "n": len(x), "sum": sum(x)}
                    tot[c] += sum(x)
passes = None
try:
    line = [ln for ln in open(os.path.join(O, "fetch.log")) if ln.startswith("{")][-1]
    d = json.loads(line)
    passes = d["warmup"] + d["steps"] + 1  # + the roofline leg's pass
    scan_evals = d["scan_evaluations_per_step_rank0"] * passes
    hbm = (2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) * 1024.0
    out["batch_passes_in_pmc_pass"] = passes
    out["scan_evaluations_in_pmc_pass"] = scan_evals
    out["derivative_kernels_hbm_bytes_in_pmc_pass"] = hbm
    out["derivative_kernels_hbm_bytes_per_scan_evaluation"] = hbm / scan_evals
    out["algorithmic_bytes_per_scan_evaluation"] = d["roofline"]["algorithmic_bytes_per_scan_evaluation"]
except Exception as e:
    out["error"] = repr(e)
json.dump(out, open(os.path.join(R, "profiles", "%s_batch64_summary.json" % tag), "w"), indent=1)
shutil.copy(os.path.join(O, "kernel_stats.csv"), os.path.join(R, "profiles", "%s_batch64_kernel_stats.csv" % tag))
print(json.dumps({k: v for k, v in out.items() if k not in ("kernel_stats", "pmc")}, indent=1))
