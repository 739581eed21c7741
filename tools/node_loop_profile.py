"""The node loop (apps/map_sequence.cpp) on a sequence of node-sized scans: per-scan wall time with the program's own per-stage
split, then the same run under rocprofv3 --kernel-trace --stats (a fresh child process) -> one JSON record with the kernel table.
    python tools/node_loop_profile.py [scans] [points per raw scan] [mode: node | rosbag] [out.json]"""
import csv, glob, json, os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from toyslam_amd import _lib, clouds, ndt
_lib.build()
n_scans = int(sys.argv[1]) if len(sys.argv) > 1 else 40
n_raw = int(sys.argv[2]) if len(sys.argv) > 2 else 60000
mode = sys.argv[3] if len(sys.argv) > 3 else "node"
out_path = sys.argv[4] if len(sys.argv) > 4 else None
rng = np.random.default_rng(3)
world = clouds.target_surfaces(4 * n_raw, seed=77, extent=60.0)[:, :3].astype(np.float32)
tmp = tempfile.mkdtemp(prefix="nodeloop_")
pose = np.eye(4)
for k in range(1, n_scans + 1):
    pose = pose @ clouds.make_T([0.3, 0.05 * np.sin(k), 0.0], np.deg2rad([0.0, 0.0, 1.0]))
    pick = world[rng.choice(len(world), n_raw, replace=False)]
    ndt.pcd_write_xyz(os.path.join(tmp, "cloud_%d.pcd" % k), (clouds.apply_T(np.linalg.inv(pose), pick) + rng.normal(0, 0.01, pick.shape)).astype(np.float32))
exe = os.path.join(tmp, "map_sequence")
libdir = os.path.join(ROOT, "toyslam_amd")
subprocess.check_call(["g++", "-std=c++17", "-O2", "-pthread", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "apps", "map_sequence.cpp"),
                       "-o", exe, "-L" + libdir, "-lndt_mi355", "-Wl,-rpath," + libdir])
leaf = "0.3" if mode == "rosbag" else "0.5"
rec = {"workload": "%d scans of %d raw points, %s m prefilter, %s-node parameters (apps/map_sequence.cpp)" % (n_scans, n_raw, leaf, mode)}
for variant, extra in (("device_resident", []), ("host_clouds", ["serial", "host"])):
    best = None
    for rep in range(3):
        r = subprocess.run([exe, tmp, leaf, "-", mode] + extra, text=True, capture_output=True)
        if r.returncode != 0:
            best = None
            rec[variant] = {"error": r.stderr[-400:]}
            break
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("time:")][0]
        total = float(line.split()[2])
        kept = int([ln for ln in r.stdout.splitlines() if ln.startswith("clouds ")][0].split()[1])
        if best is None or total < best[0]:
            best = (total, line, kept, r.stdout)
    if best:
        rec[variant] = {"total_ms": best[0], "ms_per_scan": best[0] / max(1, best[2]), "scans": best[2], "line": best[1]}
        rec[variant + "_trajectory_tail"] = [ln for ln in best[3].splitlines() if ln.startswith("  ")][-4:]
# the same run under the profiler (kernel table)
prof = os.path.join(tmp, "prof")
env = dict(os.environ, TMPDIR="/tmp")
r = subprocess.run(["rocprofv3", "--kernel-trace", "--stats", "-d", prof, "--output-format", "csv", "--", exe, tmp, leaf, "-", mode],
                   cwd="/tmp", env=env, text=True, capture_output=True)
stats = sorted(glob.glob(os.path.join(prof, "*", "*kernel_stats.csv")))
if stats:
    rows = list(csv.DictReader(open(stats[-1])))
    rec["kernels"] = [{"name": x["Name"][:90], "calls": int(x["Calls"]), "avg_us": round(float(x["AverageNs"]) / 1e3, 2),
                       "total_us_per_scan": round(float(x["TotalDurationNs"]) / 1e3 / n_scans, 2)} for x in rows[:24]]
    rec["kernel_us_per_scan"] = round(sum(float(x["TotalDurationNs"]) for x in rows) / 1e3 / n_scans, 1)
    rec["launches_per_scan"] = round(sum(int(x["Calls"]) for x in rows) / n_scans, 1)
else:
    rec["kernels"] = {"error": (r.stderr or "")[-300:]}
s = json.dumps(rec, indent=1)
print(s)
if out_path:
    open(out_path, "w").write(s + "\n")
