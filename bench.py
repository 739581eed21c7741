#!/usr/bin/env python3
"""bench.py -- scan registrations/sec of the MI355X NDT core (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload auto|single|mapbuild|batch|large|pyramid]

N = 1 (default workload "single" = BASELINE configs[1], the configuration the metric is quoted on):
  a "step" is ONE registration (pcl::Registration::align) of one synthetic 100k-point source scan against
  a 1M-point target whose voxel grid is already resident in HBM -- exactly the region
  ndt_omp/apps/align.cpp:20-29 times -- at 1.0 m voxels, DIRECT7, with the Newton loop pinned to 30 outer
  passes (max_iterations 28, transformation_epsilon 1e-9 as SURVEY 8(d) fixes the work: ndt_omp_impl.hpp:158-164
  then runs max_iterations + 2 passes).
N > 1 (default workload "mapbuild" = BASELINE configs[3]): 512 source scans of 100k points against the one
  shared 1M-point target, the scans split over the ranks by toyslam_amd.dist.shard_range, the target grid
  replicated per GPU; a "step" is one lock-step batch registration (ndt_align_batch_device) of every rank's
  share, i.e. 512 registrations.  Registrations of different scans are independent, so there is no collective
  in the data path; total work is fixed as N grows ("scaling": "strong").  After the timed region the same
  batch is run once more in the literal north_star form -- every rank steps all 512 Newton / More-Thuente
  state machines, ONE RCCL all-reduce of the zero-padded [512][32] f64 rows per lock-step, issued from C++
  on the library's stream (ndt_comm_*) -- and reported as "lockstep_allreduce".

Launch: the driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).  Started plainly with --gpus N > 1 and no
WORLD_SIZE, this file spawns the N ranks itself (child processes, before anything touches the GPU -- never an
exec) and relays rank 0's JSON line.

Rank 0 prints ONE JSON line.  Extra legs on rank 0 (outside the timed region):
  roofline     -- average duration of the dominant kernel from HIP events on the library's own stream over a
                  second pass of the same steps; algorithmic bytes = SURVEY 8(d)'s per-evaluation figure x the
                  evaluations the launch served.
  cpu_baseline -- the oracle (oracle/, a faithful OpenMP port of the reference's algorithm; the real pclomp
                  cannot be built here: no PCL / Eigen in the image) on the same inputs on the host cores, at all
                  physical cores and at 16 threads, plus an optimised CPU variant (N = 1 only).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# numpy's BLAS pool: one thread.  The synthetic inputs are made with a few (N x 3) @ (3 x 3) products; OpenBLAS starts one
# worker per visible core for them (64 on the GPU hosts) and the workers keep spinning for a while afterwards.  On a
# box whose cgroup grants 16 CPUs of a 256-thread host that spinning exhausts the CFS quota tens of milliseconds into
# the timed region, and the kernel then parks EVERY thread of the process -- the one polling for evaluation results
# included -- for 40-80 ms (measured: cpu.stat nr_throttled, and NDT_TIMING=2's longest-poll-gap).  Set before numpy loads.
for _v in ("OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ.setdefault(_v, "1")

METRIC = "scan registrations/sec (100k-pt src vs 1M-pt target, 30 Newton iters)"
MAX_ITER, EPS = 28, 1e-9  # max_iterations_ = 28, transformation_epsilon_ = 1e-9 => 30 outer passes (trap 9)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E (MI355X_MICROARCH.md)
VALU_PEAK_WAVE_INSTS_PER_S = 1024 * 2.4e9 / 4  # 256 CUs x 4 SIMDs, one wave64 VALU instruction per 4 cycles at 2.4 GHz


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--warmup", type=int, default=None)
    ap.add_argument("--set", default="U", choices=["U", "S"], help="U = uniform box (headline), S = surface scene")
    ap.add_argument("--workload", default="auto",
                    choices=["auto", "single", "mapbuild", "batch", "large", "pyramid", "selftest"],
                    help="auto = single at N = 1, mapbuild at N > 1; single = configs[1]; mapbuild = configs[3] (--scans "
                         "scans split over the ranks); batch = --batch scans per GPU; large = configs[2] (2M-pt source "
                         "vs 10M-pt target, 0.5 m voxels, --extent m scene); pyramid = configs[4] (2.0 -> 1.0 -> 0.5 m "
                         "on a streamed sequence of 2M-pt PCD scans); selftest = launcher / rendezvous check without a GPU")
    ap.add_argument("--scans", type=int, default=512, help="total scans of the mapbuild workload (configs[3]: 512)")
    ap.add_argument("--batch", type=int, default=64, help="scans per GPU for --workload batch")
    ap.add_argument("--extent", type=float, default=400.0, help="scene size of --workload large / pyramid (SURVEY 8(d): 400 m)")
    ap.add_argument("--seq-scans", type=int, default=16, help="scans of the streamed sequence of --workload pyramid")
    ap.add_argument("--near-guess", action="store_true",
                    help="--workload large: register from a guess 4 cm / 0.03 deg off T_gt with epsilon 1e-3, 35 iterations -- the "
                         "align(output, guess) path of ndt_rosbag_mapping_node.cpp:130 (tests/golden/large_golden.json: cfgB_near)")
    ap.add_argument("--lockstep-steps", type=int, default=3, help="N > 1 mapbuild: timed steps of the RCCL lock-step leg")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pmc", action="store_true", help="N = 1 single: do not run the two rocprofv3 --pmc child passes that fill roofline.traffic")
    ap.add_argument("--pmc-child", action="store_true", help="(internal) the short run a --pmc pass wraps: timed steps only, no legs")
    ap.add_argument("--no-bind", action="store_true", help="leave the CPU affinity of the rank alone (default: the cores of the GPU's NUMA node)")
    ap.add_argument("--no-mapbuild-leg", action="store_true", help="N = 1 single: skip the 512-scan map-build leg")
    ap.add_argument("--no-lockstep-leg", action="store_true", help="N > 1 mapbuild: skip the RCCL lock-step leg")
    ap.add_argument("--backend", default=None, help="torch.distributed backend (default nccl = RCCL; gloo for the selftest)")
    ap.add_argument("--dry-run", action="store_true", help="self-launch only: print the per-rank environment plan and exit")
    ap.add_argument("--selftest-die-rank", type=int, default=-1,
                    help="selftest: this rank exits with code 7 AFTER the rendezvous and the first collective (a rank lost in the middle of "
                         "a run, e.g. after ndt_comm_init_rank): the launcher must stop the others and return non-zero")
    ap.add_argument("--rehearse-one-gpu", action="store_true",
                    help="N > 1 ranks that all use GPU 0 and rendezvous over gloo: exercises the sharded workload, the launcher and the "
                         "JSON line on a one-GPU box (RCCL refuses two ranks on one device, so the lock-step leg reports an error)")
    return ap.parse_args(argv)


# =====================================================================================================
# self-launch: python bench.py --gpus N with no WORLD_SIZE -> N child ranks (never an exec: a process
# that has touched the GPU must not be replaced, and the parent never touches it)
# =====================================================================================================
def rank_environments(n, port, base_env=None):
    envs = []
    for r in range(n):
        e = dict(os.environ if base_env is None else base_env)
        e.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                  "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool (RCCL needs it)
        e.setdefault("OMP_NUM_THREADS", "4")
        envs.append(e)
    return envs


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(args, argv):
    port = free_port()
    envs = rank_environments(args.gpus, port)
    if args.dry_run:
        plan = [{k: e[k] for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")} for e in envs]
        print(json.dumps({"launcher": "self", "n_ranks": args.gpus, "ranks": plan,
                          "command": [sys.executable, os.path.abspath(__file__)] + argv}))
        return 0
    import tempfile
    procs = []
    out0 = tempfile.TemporaryFile()  # rank 0's stdout (a pipe nobody drains while we poll could fill up)
    for r, e in enumerate(envs):
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL, stderr=None))
    rcs = supervise(procs, float(os.environ.get("NDT_BENCH_LAUNCH_BUDGET_S", "1500")))
    out0.seek(0)
    lines = [ln for ln in out0.read().decode("utf-8", "replace").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    rc = max(abs(c) for c in rcs)
    if rc or not lines:
        sys.stderr.write("bench.py self-launch: rank exit codes %s\n" % rcs)
    return rc if rc else (0 if lines else 1)


def supervise(procs, budget_s, grace_s=20.0, poll_s=0.2):
    """Wait for every child.  If one exits non-zero (a rank that dies before the rendezvous leaves the others waiting in
    init_process_group or in a collective for ever) or the wall-clock budget runs out, the remaining ranks get SIGTERM,
    then SIGKILL after grace_s.  Returns the exit codes (a killed child reports its negative signal number; a child
    stopped because of ANOTHER child's failure counts as failed too).  Never execs, never kills by pattern: only the
    PIDs started here."""
    t0 = time.monotonic()
    failed = False
    while True:
        rcs = [p.poll() for p in procs]
        if all(c is not None for c in rcs):
            break
        if any(c not in (None, 0) for c in rcs) or time.monotonic() - t0 > budget_s:
            failed = True
            break
        time.sleep(poll_s)
    if failed:
        why = "budget of %.0f s exhausted" % budget_s if all(c in (None, 0) for c in rcs) else "a rank failed"
        sys.stderr.write("bench.py self-launch: %s (exit codes so far %s); stopping the remaining ranks\n" % (why, rcs))
        for p in procs:
            if p.poll() is None:
                p.terminate()
        t1 = time.monotonic()
        while any(p.poll() is None for p in procs) and time.monotonic() - t1 < grace_s:
            time.sleep(poll_s)
        for p in procs:
            if p.poll() is None:
                p.kill()
        for p in procs:
            p.wait()
        rcs = [p.returncode if p.returncode else 1 for p in procs]
    return rcs


# =====================================================================================================
# helpers
# =====================================================================================================
def algorithmic_bytes_per_eval(n_points, mean_neighbors):
    """16 B point + 7 x 4 B voxel-slot probes + 36 B per valid neighbour record."""
    return n_points * (16 + 7 * 4 + 36 * mean_neighbors)


def host_info():
    """CPU model string, logical CPUs this process may use, physical cores among them."""
    model = None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    cpus = sorted(os.sched_getaffinity(0))
    cores = set()
    for c in cpus:
        try:
            pkg = open("/sys/devices/system/cpu/cpu%d/topology/physical_package_id" % c).read().strip()
            core = open("/sys/devices/system/cpu/cpu%d/topology/core_id" % c).read().strip()
            cores.add((pkg, core))
        except OSError:
            cores.add(("?", str(c)))
    quota = None  # cgroup v2 CPU bandwidth: "max" or "<quota_us> <period_us>" -- what the box really grants
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    granted = len(cores) if quota is None else max(1, min(len(cores), int(quota)))  # threads the kernel will really run
    return {"cpu_model": model, "logical_cpus": len(cpus), "physical_cores": len(cores), "cgroup_cpu_quota": quota,
            "granted_cpus": granted}


def cgroup_throttle_counters():
    """(nr_throttled, throttled_usec) of this process's cgroup (v2 cpu.stat), or None: how often the kernel parked the
    cgroup's threads for having used up its CPU bandwidth -- the failure mode of too many spinning host threads."""
    try:
        d = dict(ln.split() for ln in open("/sys/fs/cgroup/cpu.stat").read().splitlines() if ln.strip())
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return None


def bind_near_gpu(local_rank):
    """Keep this rank's host thread (it polls pinned memory once per evaluation) and its pinned
    allocations on the NUMA node the GPU hangs off.  Best effort: returns a description or None."""
    try:
        import torch
        pr = torch.cuda.get_device_properties(local_rank)
        bdf = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
        node = int(open("/sys/bus/pci/devices/%s/numa_node" % bdf).read())
        if node < 0:
            return None
        cpus = set()
        for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus.update(range(int(lo), int(hi or lo) + 1))
        cpus &= os.sched_getaffinity(0)
        if len(cpus) < 4:
            return None
        os.sched_setaffinity(0, cpus)
        return {"gpu_pci": bdf, "numa_node": node, "cpus": len(cpus)}
    except Exception:
        return None


def committed_profile(name):
    """Newest committed profiles/rNN_<name> (PMC passes cannot run inside this process: they need rocprofv3)."""
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + name)))
    if not c:
        return None, None
    try:
        return json.load(open(c[-1])), os.path.relpath(c[-1], ROOT)
    except Exception:
        return None, None


def pmc_traffic_of_headline_kernel(set_name):
    """HBM traffic of the headline kernel measured by THIS run: two fresh child processes -- started before this process touches the
    GPU, the program directly behind `--` -- run a short pass of the same workload under `rocprofv3 --kernel-trace --pmc X`,
    FETCH_SIZE and WRITE_SIZE in passes of their own (MI355X_MICROARCH.md: KB, 2 x FETCH_SIZE on gfx950).
    -> (bytes per launch or None, dict of details)."""
    import csv
    import glob
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if not exe:
        return None, {"skipped": "rocprofv3 not found"}
    vals, info = {}, {"passes": {}}
    t0 = time.perf_counter()
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="ndt_pmc_")
        cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable, os.path.abspath(__file__),
               "--pmc-child", "--set", set_name, "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-mapbuild-leg", "--no-bind"]
        try:
            r = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), capture_output=True, text=True, timeout=170)
        except Exception as e:
            shutil.rmtree(d, ignore_errors=True)
            return None, {"skipped": "pass %s: %r" % (counter, e)}
        per = []
        for cf in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(cf)):
                if "k_eval_server" in row.get("Kernel_Name", "") and row.get("Counter_Name") == counter:
                    per.append(float(row["Counter_Value"]))
        shutil.rmtree(d, ignore_errors=True)
        if not per:
            return None, {"skipped": "pass %s: no k_eval_server rows (rc %d): %s" % (counter, r.returncode, (r.stderr or "")[-200:])}
        per = per[2:] if len(per) > 4 else per  # (the first launches of a process fetch code and cold tables)
        vals[counter] = sum(per) / len(per)
        info["passes"][counter] = {"launches": len(per), "mean_kb": vals[counter]}
    info["seconds"] = time.perf_counter() - t0
    info["how"] = ("this run's own counters: two child processes under rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes), "
                   "mean over the warm k_eval_server launches, (2 x FETCH_SIZE + WRITE_SIZE) x 1024 bytes")
    return (2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0, info


def thread_cpu_times():
    """{tid: (comm, user+system seconds)} of this process's threads (/proc/self/task)."""
    out = {}
    tck = os.sysconf("SC_CLK_TCK")
    for t in os.listdir("/proc/self/task"):
        try:
            f = open("/proc/self/task/%s/stat" % t).read()
            comm = f[f.index("(") + 1:f.rindex(")")]
            rest = f[f.rindex(")") + 2:].split()
            out[int(t)] = (comm, (int(rest[11]) + int(rest[12])) / tck)
        except Exception:
            pass
    return out


def median_time(fn, n=5):
    import numpy as np
    ts = []
    for _ in range(n):
        ta = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - ta)
    return float(np.median(ts))


def timed_oracle(po, tgt, src, resolution, threads, optimised, budget_s):
    """The oracle's align() on the workload: 1 warm-up (also the parity reference) + up to 5 timed runs within budget_s."""
    import numpy as np
    o = po.OracleNDT(resolution=resolution, search_method=po.DIRECT7, num_threads=threads, trans_eps=EPS, max_iter=MAX_ITER,
                     optimised=optimised)
    tb = time.perf_counter()
    o.set_target(tgt)
    tb = time.perf_counter() - tb
    o.set_source(src)
    r = o.align()
    times = []
    deadline = time.perf_counter() + budget_s
    while len(times) < 5 and (not times or time.perf_counter() < deadline):
        ta = time.perf_counter()
        o.align()
        times.append(time.perf_counter() - ta)
    med = float(np.median(times))
    return med, len(times), tb, r


def near_T_gt(T, T_gt, rot=2e-3, trans=2e-2):
    """Did a registration end at the known transform?  (set U carries 2 cm of noise per point and no structure but its
    sampling: the estimate lands within millimetres, not at the oracle-parity tolerance.)"""
    import numpy as np
    return bool(np.abs(T[:3, :3] - T_gt[:3, :3]).max() < rot and np.abs(T[:3, 3] - T_gt[:3, 3]).max() < trans)


def mapbuild_scans(clouds, tgt, lo, hi, n_source):
    """Scans lo..hi-1 of the configs[3] workload (clouds.mapbuild_scan: seeds depend on the scan number only)."""
    scans, T_gts = [], []
    for k in range(lo, hi):
        s, T = clouds.mapbuild_scan(tgt, k, n_source)
        scans.append(s)
        T_gts.append(T)
    return scans, T_gts


# =====================================================================================================
def main():
    argv = sys.argv[1:]
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, argv))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    workload = args.workload
    if workload == "auto":
        workload = "single" if world == 1 else "mapbuild"
    # (single: 200 steps = ~90 ms of timed region; round 3's 20-50 steps were 9-23 ms, thin for a +-3 % claim)
    steps = args.steps if args.steps is not None else {"single": 200, "large": 10, "mapbuild": 3, "batch": 5, "pyramid": 5, "selftest": 3}[workload]
    warmup = args.warmup if args.warmup is not None else {"single": 10, "large": 2, "mapbuild": 1, "batch": 1, "pyramid": 2, "selftest": 0}[workload]

    # N = 1 headline: the counter passes run first, in child processes, while this process has not touched the GPU yet
    pmc = None
    if world == 1 and workload == "single" and not args.pmc_child and not args.no_pmc:
        try:
            pmc = pmc_traffic_of_headline_kernel(args.set)
        except Exception as e:  # never lose the headline line to an auxiliary leg
            pmc = (None, {"skipped": repr(e)})

    import numpy as np
    import torch
    dist = None
    use_gpu = workload != "selftest"
    if args.rehearse_one_gpu:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        backend = args.backend or ("gloo" if (args.rehearse_one_gpu or not use_gpu) else "nccl")
        if use_gpu:
            torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    elif use_gpu and torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    def barrier():
        if use_gpu and torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    def max_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if (use_gpu and dist.get_backend() == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cuda" if (use_gpu and dist.get_backend() == "nccl") else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t.item())

    from toyslam_amd import dist as nd

    # ---- launcher / rendezvous self-test (CPU, gloo): what a SCALE run exercises before any kernel ----
    if workload == "selftest":
        lo, hi = nd.shard_range(args.scans, rank, world)
        barrier()
        if args.selftest_die_rank == rank:
            os._exit(7)  # (the others are on their way into the next collective)
        t0 = time.perf_counter()
        for _ in range(steps):
            time.sleep(0.001 * (hi - lo) / max(1, args.scans) * world)
        barrier()
        dt = max_over_ranks(time.perf_counter() - t0)
        covered = sum_over_ranks(hi - lo)
        if rank == 0:
            print(json.dumps({"metric": "launcher selftest", "value": steps * args.scans / dt, "unit": "scans/s", "n_gpus": 0,
                              "world_size": world, "steps": steps, "warmup": warmup, "scans_covered": covered,
                              "backend": dist.get_backend() if dist is not None else None}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    binding = bind_near_gpu(local_rank) if (torch.cuda.is_available() and not args.no_bind) else None
    from toyslam_amd import clouds, ndt

    # ---- inputs (synthetic, seeded; resident in HBM before the timed region) ----
    M_TARGET, N_SOURCE, RESOLUTION = 1000000, 100000, 1.0
    if workload in ("large", "pyramid"):
        M_TARGET, N_SOURCE, RESOLUTION = 10000000, 2000000, 0.5
        # the set-S generator scaled to 400 x 400 m (--extent; round 1 measured a 200 m scene with
        # 120 boxes, kept as --extent 200)
        tgt = clouds.target_surfaces(M_TARGET, extent=args.extent, n_boxes=60 if args.extent >= 300 else 120)
    else:
        tgt = clouds.target_uniform(M_TARGET) if args.set == "U" else clouds.target_surfaces(M_TARGET)

    if workload == "pyramid":
        out = run_pyramid(args, ndt, clouds, tgt, local_rank, steps, warmup, binding, world, rank, barrier, max_over_ranks)
        if rank == 0:
            print(json.dumps(out), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    reg = ndt.NormalDistributionsTransform(device=local_rank)
    reg.setResolution(RESOLUTION)
    reg.setNeighborhoodSearchMethod(ndt.DIRECT7)
    near = workload == "large" and args.near_guess
    guess = None
    if near:  # oracle/gen_golden_large.py: NEAR_GUESS, cfgB_near
        guess = clouds.make_T([0.30 + 0.04, -0.20 - 0.03, 0.10 + 0.02], np.deg2rad([0.5 + 0.03, -0.3 - 0.02, 1.0 + 0.04]))
        reg.setMaximumIterations(35)
        reg.setTransformationEpsilon(1e-3)
    else:
        reg.setMaximumIterations(MAX_ITER)
        reg.setTransformationEpsilon(EPS)
    t0 = time.perf_counter()
    reg.setInputTarget(tgt)
    t_build_first = time.perf_counter() - t0
    t_build = t_build_dev = t_build_ref = t_build_cloud = t_build_first
    if not args.pmc_child:
      t_build = median_time(lambda: reg.setInputTarget(tgt))                 # host buffer: H2D over PCIe + grid build
      tgt_dev = torch.from_numpy(np.c_[tgt, np.ones(len(tgt), np.float32)]).cuda()
      torch.cuda.synchronize()
      t_build_dev = median_time(lambda: reg.setInputTargetDevice(tgt_dev.data_ptr(), len(tgt), 16))  # cloud already in HBM, copied
      # ... and used where it lies (ndt_set_input_target_device_ref: pcl::Registration keeps a shared pointer, it never copies either)
      t_build_ref = median_time(lambda: reg.setInputTargetDeviceRef(tgt_dev.data_ptr(), len(tgt)))
      # ... and from an ndt_cloud (a cloud the library made or uploaded itself keeps its bounding boxes: no box pass, no host round trip)
      tgt_cloud = reg.uploadCloud(tgt)
      # (nothing in this call waits for the device -- no box to poll for -- so it is timed as eight builds back to back and one
      # synchronisation, the way the by-reference builds above pace themselves through their polled boxes)
      def cloud_builds():
          for _ in range(8):
              reg.setInputTargetCloud(tgt_cloud)
          torch.cuda.synchronize()
      t_build_cloud = median_time(cloud_builds) / 8
      reg.setInputTargetDeviceRef(tgt_dev.data_ptr(), len(tgt))  # (the timed region runs against the by-reference grid, as in round 3)
      tgt_cloud.release()

    T_gts = None
    if workload in ("single", "large"):
        src = clouds.source_from_target(tgt, N_SOURCE, seed=clouds.SEED + 1 + 7 * rank)
        reg.setInputSource(src)
        t_source = median_time(lambda: reg.setInputSource(src))           # H2D + spatial ordering of the scan

        def step():
            reg.align(guess)
        regs_per_step_global = world  # replicas: every rank registers its own scan
        scaling = "weak"
    else:
        if workload == "mapbuild":
            lo, hi = nd.shard_range(args.scans, rank, world)
            regs_per_step_global = args.scans
            scaling = "strong"
        else:
            lo, hi = rank * args.batch, (rank + 1) * args.batch
            regs_per_step_global = args.batch * world
            scaling = "weak"
        scans, T_gts = mapbuild_scans(clouds, tgt, lo, hi, N_SOURCE)
        n_local = hi - lo
        cat = np.ascontiguousarray(np.concatenate(scans, axis=0)) if scans else np.zeros((0, 3), np.float32)
        del scans
        dev = torch.from_numpy(np.c_[cat, np.ones(len(cat), np.float32)]).cuda()
        del cat
        offsets = np.arange(n_local + 1, dtype=np.uintp) * N_SOURCE
        last = {}

        def step():
            if n_local:
                last["res"] = reg.alignBatch(device_ptr=dev.data_ptr(), offsets=offsets, stride_bytes=16)

    for _ in range(warmup):
        step()
    barrier()
    thr0 = thread_cpu_times() if os.environ.get("NDT_BENCH_THREADS") else None
    cg0 = cgroup_throttle_counters()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier()
    dt_local = time.perf_counter() - t0
    cg1 = cgroup_throttle_counters()
    if thr0 is not None:  # diagnostics: which host threads burned CPU during the timed region (cgroup quota hunting)
        thr1 = thread_cpu_times()
        used = sorted(((thr1[t][1] - thr0.get(t, (None, 0))[1], thr1[t][0], t) for t in thr1), reverse=True)
        sys.stderr.write("[bench threads] %d threads, timed region %.3f s; busiest (cpu-s, comm, tid): %s\n" % (len(thr1), dt_local, used[:12]))
    dt = max_over_ranks(dt_local)
    if args.pmc_child:  # the profiler around this process has what it came for
        print(json.dumps({"pmc_child": True, "steps": steps, "ms_per_step": dt / steps * 1e3}), flush=True)
        return

    # every rank: did its registrations end at the known T_gt?  (summed over ranks; outside the timed region)
    recovered = None
    if T_gts is not None:
        ok = 0
        if T_gts:
            Ts = last["res"]["T"]
            for k, Tg in enumerate(T_gts):
                ok += int(near_T_gt(Ts[k], Tg))
        recovered = int(sum_over_ranks(ok))
    per_rank_regs = None
    if dist is not None:
        mine = torch.zeros(world, dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        mine[rank] = (steps * (len(T_gts) if T_gts is not None else 1)) / dt_local
        dist.all_reduce(mine)
        per_rank_regs = [float(x) for x in mine.tolist()]

    out = None
    if rank == 0:
        value = steps * regs_per_step_global / dt
        names = {
            "single": "single 100k-pt source vs 1M-pt target, 1.0 m voxels, DIRECT7, 30 Newton passes (max_iterations 28, epsilon 1e-9), set " + args.set,
            "large": ("single 2M-pt source vs 10M-pt target (surface scene, %g m), 0.5 m voxels, DIRECT7, " % args.extent) +
                     ("from a guess 4 cm / 0.03 deg off T_gt, epsilon 1e-3, max_iterations 35 (ndt_rosbag_mapping_node.cpp:130)" if near else "30 Newton passes"),
            "mapbuild": "map-build: %d x 100k-pt sources vs one shared 1M-pt target, scans split over %d GPU(s), lock-step batch per GPU, set %s" % (args.scans, world, args.set),
            "batch": "map-build batch: %d x 100k-pt sources per GPU vs one 1M-pt target, lock-step, set %s" % (args.batch, args.set),
        }
        out = {
            "metric": METRIC, "value": value, "unit": "registrations/s", "n_gpus": world, "steps": steps,
            "warmup": warmup, "ms_per_step": dt / steps * 1e3, "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": names[workload], "target_points": M_TARGET, "source_points": N_SOURCE, "resolution_m": RESOLUTION,
                       "search": "DIRECT7", "outer_passes": (None if near else MAX_ITER + 2),
                       "sharding": ("scans split over the ranks (dist.shard_range), target grid replicated, no collective in the data path"
                                    if workload in ("mapbuild", "batch") else "one scan stream per GPU, target grid replicated")},
            "world_size": world, "collective_backend": (dist.get_backend() if dist is not None else None),
            "per_rank_registrations_per_s": per_rank_regs,
            "cgroup_nr_throttled_in_timed_region_rank0": (cg1[0] - cg0[0]) if (cg0 and cg1) else None,
            "host_binding": binding, "target_build_ms": t_build * 1e3,
            # (key definitions as in rounds 1-2: device_resident = ndt_set_input_target_device, the library copies the cloud;
            #  the by-reference build and the build from an ndt_cloud have keys of their own)
            "target_build_roofline": None, "target_build_device_resident_ms": t_build_dev * 1e3,
            "target_build_device_resident_copied_ms": t_build_dev * 1e3,
            "target_build_device_ref_ms": t_build_ref * 1e3,
            "target_build_resident_cloud_ms": t_build_cloud * 1e3,
            "target_build_first_call_ms": t_build_first * 1e3,
        }
        if recovered is not None:
            out["registrations_ending_at_T_gt"] = recovered
            out["registrations_per_step"] = regs_per_step_global
        try:  # K1 against its own roof (M x 16 B read + V x 64 B of records written)
            gi = reg.grid_counts()
            k1_bytes = M_TARGET * 16 + gi["n_leaves"] * 64
            out["target_build_roofline"] = {"bound": "hbm", "algorithmic_bytes": k1_bytes,
                                            "achieved": k1_bytes / t_build_ref / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": k1_bytes / t_build_ref / 1e9 / HBM_PEAK_GBS,
                                            "occupied_voxels": gi["n_leaves"], "valid_voxels": gi["n_valid"],
                                            "frac_resident_cloud": k1_bytes / t_build_cloud / 1e9 / HBM_PEAK_GBS,
                                            "note": "wall time of ndt_set_input_target_device_ref (cloud already in HBM and used where it lies), all of its kernels "
                                                    "(target_build_device_ref_ms); frac_resident_cloud: the same from an ndt_cloud, whose boxes are known "
                                                    "(target_build_resident_cloud_ms); target_build_device_resident_ms: with the library's own copy of the cloud"}
        except Exception:
            pass

    if workload in ("single", "large"):
        if rank == 0:
            single_legs(out, args, reg, ndt, clouds, tgt, src, workload, world, steps, dt, t_build, t_source, N_SOURCE, RESOLUTION, value, guess)
            if pmc is not None and "roofline" in out:
                out["roofline"]["traffic"] = pmc[0]
                out["roofline"]["traffic_measured"] = pmc[1]
                if pmc[0]:
                    out["roofline"]["traffic_over_algorithmic"] = pmc[0] / out["roofline"]["algorithmic_bytes_per_launch"]
    else:
        batch_legs(out, args, reg, nd, ndt, dist, rank, world, workload, step, steps, dt, N_SOURCE, dev if n_local else None, offsets, lo, hi,
                   T_gts, torch)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


# =====================================================================================================
# N1 / N2 at the mapping nodes' size: a 60 k-point raw scan of a 60 m scene, 0.5 m leaf; clouds resident in HBM
# =====================================================================================================
def node_rows(reg, ndt, clouds, np):
    """pcl::VoxelGrid prefilter (N1) and update_global_map (N2) against their own roof: M x 16 B read + V x 16 B written
    (SURVEY 8(f)); wall time of the call (a chain of small kernels, the count and the result's boxes come back to the host)."""
    import torch
    n_raw, leaf = 60000, 0.5
    rng = np.random.default_rng(3)
    world_pts = clouds.target_surfaces(4 * n_raw, seed=77, extent=60.0)[:, :3].astype(np.float32)
    h = ndt.NormalDistributionsTransform(device=getattr(reg, "_device", 0))
    h.warmUp(65536)
    raws = [np.ascontiguousarray(np.c_[world_pts[rng.choice(len(world_pts), n_raw, replace=False)] + [0.3 * k, 0.0, 0.0], np.ones(n_raw)].astype(np.float32)) for k in range(8)]
    devs = [torch.from_numpy(r).cuda() for r in raws]
    torch.cuda.synchronize()

    def med(f, n):
        ts = []
        for i in range(n):
            t0 = time.perf_counter()
            f(i)
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts))
    kept = {}

    def n1(i):
        c, _ = h.voxelGridFilterCloudDevice(devs[i % 8].data_ptr(), n_raw, 16, leaf)
        kept["n"] = len(c)
        kept["c"] = c
    t_n1 = med(n1, 24)
    t_n1_host = med(lambda i: h.voxelGridFilterCloud(raws[i % 8], leaf)[0].release(), 16)
    v1 = kept["n"]
    # N2: the map after 20 scans, then the update by one more filtered scan
    h.mapClear()
    filt = [h.voxelGridFilterCloud(r, leaf)[0] for r in raws]
    for k in range(20):
        h.mapUpdateCloud(filt[k % 8], clouds.make_T([0.3 * k, 0.02 * k, 0.0], [0.0, 0.0, 0.01 * k]), 0.5)
    m_before = h.mapSize()
    sizes = []

    def n2(i):
        sizes.append(h.mapSize())
        h.mapUpdateCloud(filt[i % 8], clouds.make_T([0.3 * (20 + i), 0.02 * i, 0.0], [0.0, 0.0, 0.01 * i]), 0.5)
    t_n2 = med(n2, 16)
    m_in = float(np.median(sizes)) + v1
    v2 = h.mapSize()
    b1 = n_raw * 16 + v1 * 16
    b2 = m_in * 16 + v2 * 16
    # ... and N1 on a 1 M-point scan of the same scene (dense clouds from 128 k points take K1's bucket front end)
    big = {}
    try:
        n_big = 1000000
        wb = clouds.target_surfaces(n_big, seed=78, extent=60.0)[:, :3].astype(np.float32)
        db = torch.from_numpy(np.ascontiguousarray(np.c_[wb, np.ones(n_big)].astype(np.float32))).cuda()
        torch.cuda.synchronize()
        kb = {}

        def n1b(i):
            c, _ = h.voxelGridFilterCloudDevice(db.data_ptr(), n_big, 16, leaf)
            kb["n"] = len(c)
        t_b = med(n1b, 12)
        bb = n_big * 16 + kb["n"] * 16
        big = {"us_per_call": t_b * 1e6, "points_in": n_big, "voxels_out": kb["n"], "algorithmic_bytes": bb,
               "roofline": {"bound": "hbm", "achieved": bb / t_b / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bb / t_b / 1e9 / HBM_PEAK_GBS},
               "what": "ndt_cloud_voxel_filter of a 1 M-point scan in HBM: k1_hist / k1_colscan / k1_scatter / vf_finalize / vf_bitmap_prefix / vf_place"}
        del db
    except Exception as e:
        big = {"error": repr(e)}
    return {"workload": "raw scan of %d points over a 60 m scene, %.1f m leaf; map of ~%d points" % (n_raw, leaf, m_before),
            "n1_voxel_filter": {"us_per_call": t_n1 * 1e6, "points_in": n_raw, "voxels_out": v1, "algorithmic_bytes": b1,
                                "roofline": {"bound": "hbm", "achieved": b1 / t_n1 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b1 / t_n1 / 1e9 / HBM_PEAK_GBS},
                                "from_host_buffer_us_per_call": t_n1_host * 1e6,
                                "what": "ndt_cloud_voxel_filter, input in HBM (from_host_buffer: the same from pageable host memory, upload included)"},
            "n2_map_update": {"us_per_call": t_n2 * 1e6, "points_in": m_in, "voxels_out": v2, "algorithmic_bytes": b2,
                              "roofline": {"bound": "hbm", "achieved": b2 / t_n2 / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": b2 / t_n2 / 1e9 / HBM_PEAK_GBS},
                              "what": "ndt_map_update_cloud: transform of the scan into the map's tail + voxel filter of map and scan"},
            "n1_voxel_filter_1m_points": big,
            "note": "at the nodes' size both are chains of ~10 small launches paced by launch and completion latencies, not by bytes: the fractions say so"}


# =====================================================================================================
# legs of the single-scan workloads (rank 0, outside the timed region)
# =====================================================================================================
def single_legs(out, args, reg, ndt, clouds, tgt, src, workload, world, steps, dt, t_build, t_source, N_SOURCE, RESOLUTION, value, guess=None):
    import numpy as np
    st = reg.stats()
    out["outer_iterations"] = reg.getFinalNumIteration()
    out["converged"] = bool(reg.hasConverged())
    T_timed = reg.getFinalTransformation()
    it_timed = reg.getFinalNumIteration()
    out["set_source_ms"] = t_source * 1e3
    out["registrations_per_s_incl_target_build_and_source_upload"] = 1.0 / (dt / steps + t_build + t_source)
    out["evaluations_per_registration"] = st["n_evals"]
    out["f64_hessian_recomputes"] = st["n_hessian_recomputes"]
    out["mean_neighbors"] = st["mean_neighbors"]
    out["final_vs_T_gt"] = {"rot_max_abs": float(np.abs(T_timed[:3, :3] - clouds.T_GT_DEFAULT[:3, :3]).max()),
                            "trans_max_abs_m": float(np.abs(T_timed[:3, 3] - clouds.T_GT_DEFAULT[:3, 3]).max())}
    # ---- roofline leg: HIP events on the library's stream, second pass of the same steps ----
    # The kernel of the timed region is k_eval_server: ONE launch per registration that serves
    # every evaluation of it.  ndt_profile_enable(2) brackets that launch with an event pair.
    # (three passes of up to ten registrations, the fastest pass kept: the kernel waits for the host between evaluations, so
    # a host thread that loses its CPU for a millisecond stretches one launch -- and with it a ten-launch mean -- by as much)
    n_rep = min(steps, 10)
    reg.profile(2)
    n_launch, ms, pass_us = 0, 0.0, []
    for _ in range(3):
        reg.profile_read(3)
        for _ in range(n_rep):
            reg.align(guess)
        n_i, ms_i = reg.profile_read(3)
        pass_us.append(ms_i * 1e3 / max(n_i, 1))
        if n_launch == 0 or (n_i > 0 and ms_i / n_i < ms / n_launch):
            n_launch, ms = n_i, ms_i
    st2 = reg.stats()
    # ... and, for reference, the same device code as one launch per evaluation (profile mode 1)
    reg.profile(1)
    reg.profile_read(0)
    for _ in range(n_rep):
        reg.align(guess)
    n_eval_launch, ms_eval = reg.profile_read(0)
    reg.profile(0)
    hbar = st2["mean_neighbors"]
    avg_s = ms * 1e-3 / max(n_launch, 1)
    bytes_per_eval = algorithmic_bytes_per_eval(N_SOURCE, hbar)
    evals_per_launch = st2["n_evals"] + st2["n_hessian_recomputes"]
    bytes_per_launch = evals_per_launch * bytes_per_eval
    server_used = n_launch > 0 and avg_s > 0
    if not server_used:  # NDT_PERSISTENT=0: the launch path ran; its per-evaluation kernel is the dominant one
        avg_s = ms_eval * 1e-3 / max(n_eval_launch, 1)
        evals_per_launch, bytes_per_launch, n_launch = 1, bytes_per_eval, n_eval_launch
    achieved = bytes_per_launch / avg_s / 1e9
    # HBM traffic and VALU issue of the same kernel: PMC counters need rocprofv3 around the process, so they cannot be
    # collected inside this run; the figures of the newest committed profile of this command are attached under
    # *_committed_profile keys and `traffic` stays null.
    committed = {}
    if workload == "single":
        prof, src_name = committed_profile("profile_summary.json")
        try:
            for name, c in (prof or {}).get("pmc", {}).items():
                if "k_eval_server<7>" in name and "FETCH_SIZE" in c:
                    committed["traffic_committed_profile"] = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
                    committed["traffic_committed_profile_source"] = src_name + " (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE passes, KB, 2 x FETCH_SIZE on gfx950)"
        except Exception:
            pass
        valu, valu_name = committed_profile("pmc_valu.json")
        try:
            per_wave = valu["server_valu_insts_per_wave_per_evaluation"]
            waves = [v["SQ_WAVES"] for k, v in valu["kernels"].items() if "k_eval_server<7>" in k][0]
            insts_per_launch = per_wave * waves * evals_per_launch
            committed["valu_issue_frac_committed_profile"] = insts_per_launch / avg_s / VALU_PEAK_WAVE_INSTS_PER_S
            committed["valu_insts_per_wave_per_evaluation_committed_profile"] = per_wave
            committed["valu_issue_frac_source"] = valu_name + (" (SQ_INSTS_VALU per wave per evaluation x waves x evaluations of this run / this run's "
                                                               "kernel time / (1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction))")
        except Exception:
            pass
    out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                       "kernel": ("k_eval_server<DIRECT7> (persistent: all evaluations of one registration, "
                                  "their reductions, the f64 Hessian recompute and the output transform)") if server_used else
                                 "k_derivatives_fused<DIRECT7> (one launch per evaluation: the launch path -- the library's choice for scans of "
                                 "1.5M points and more, or NDT_PERSISTENT=0)",
                       "avg_kernel_us": avg_s * 1e6, "launches_timed": n_launch,
                       "evaluations_per_launch": evals_per_launch,
                       "algorithmic_bytes_per_evaluation": bytes_per_eval,
                       "algorithmic_bytes_per_launch": bytes_per_launch, "mean_neighbors": hbar,
                       "how": "one hipEvent pair on the library stream around the kernel of each registration "
                              "(ndt_profile_enable(2)), in three more passes of the same steps; the fastest pass's mean",
                       "avg_kernel_us_by_pass": pass_us,
                       "per_evaluation_kernel": {
                           "kernel": "k_derivatives_fused<DIRECT7, hessian>: the same device code as one launch "
                                     "per evaluation (ndt_profile_enable(1))",
                           "avg_event_us": ms_eval * 1e3 / max(n_eval_launch, 1), "launches_timed": n_eval_launch,
                           "achieved_GBs": bytes_per_eval / (ms_eval * 1e-3 / max(n_eval_launch, 1)) / 1e9}}
    out["roofline"].update(committed)
    out["us_per_evaluation_in_timed_region"] = dt / steps / max(st["n_evals"] + st["n_hessian_recomputes"], 1) * 1e6
    # where an evaluation's time goes through the persistent server: a round with no per-point body (command -> every block
    # -> row stores -> shard tickets -> part sums -> host) against a full with-Hessian round (ndt_diag_server_roundtrip)
    if server_used and workload == "single":
        try:
            rt = reg.diag_server_roundtrip(ndt.host_matrix_to_pose(T_timed), 300)
            out["protocol_us_per_evaluation"] = rt["nop_us"]
            out["body_us_per_evaluation"] = rt["with_hessian_us"] - rt["nop_us"]
            out["server_round_us"] = {"no_body": rt["nop_us"], "without_hessian": rt["no_hessian_us"], "with_hessian": rt["with_hessian_us"],
                                      "how": "ndt_diag_server_roundtrip: 300 rounds of each kind through the evaluation server at the final pose"}
        except Exception as e:
            out["server_round_us"] = {"error": repr(e)}
    out["registration_algorithmic_GBs"] = (st["n_evals"] + st["n_hessian_recomputes"]) * bytes_per_eval / (dt / steps) / 1e9

    # ---- the adapter's path: pcl::Registration::align fills the caller's output cloud (include/pclomp/ndt_omp.h passes a host
    # buffer to ndt_align): the same registration plus the transformed scan's trip to host memory ----
    if workload == "single":
        try:
            n_ad = max(5, min(steps, 50))
            reg.align(guess, n_out=N_SOURCE)
            ta = time.perf_counter()
            for _ in range(n_ad):
                reg.align(guess, n_out=N_SOURCE)
            t_ad = (time.perf_counter() - ta) / n_ad
            out["adapter_path"] = {"align_with_cloud_to_host_registrations_per_s": 1.0 / t_ad, "ms_per_registration": t_ad * 1e3, "steps": n_ad,
                                   "what": "ndt_align with a host out_cloud (what pclomp::NormalDistributionsTransform::computeTransformation "
                                           "calls): the headline's registration + fused transform of the scan + %d x 16 B to host memory" % N_SOURCE}
        except Exception as e:
            out["adapter_path"] = {"error": repr(e)}
    # ---- the rows around the path at the mapping nodes' size, clouds resident in HBM ----
    if world == 1 and workload == "single":
        try:
            out["node_rows"] = node_rows(reg, ndt, clouds, np)
        except Exception as e:
            out["node_rows"] = {"error": repr(e)}

    # ---- configs[3] on this one GPU: the denominator of the 8-GPU map-build comparison ----
    if world == 1 and workload == "single" and not args.no_mapbuild_leg:
        try:
            import torch
            scans, T_gts = mapbuild_scans(clouds, tgt, 0, args.scans, N_SOURCE)
            cat = np.ascontiguousarray(np.concatenate(scans, axis=0))
            del scans
            dev = torch.from_numpy(np.c_[cat, np.ones(len(cat), np.float32)]).cuda()
            del cat
            offsets = np.arange(args.scans + 1, dtype=np.uintp) * N_SOURCE
            res = reg.alignBatch(device_ptr=dev.data_ptr(), offsets=offsets, stride_bytes=16)  # warm-up
            torch.cuda.synchronize()
            ta = time.perf_counter()
            n_mb = 2
            for _ in range(n_mb):
                res = reg.alignBatch(device_ptr=dev.data_ptr(), offsets=offsets, stride_bytes=16)
            torch.cuda.synchronize()
            tm = (time.perf_counter() - ta) / n_mb
            ok = sum(int(near_T_gt(res["T"][k], T_gts[k])) for k in range(args.scans))
            out["mapbuild_1gpu"] = {"workload": "configs[3] on one GPU: %d x 100k-pt sources vs the same target, one lock-step batch" % args.scans,
                                    "value": args.scans / tm, "unit": "registrations/s", "ms_per_step": tm * 1e3, "steps": n_mb,
                                    "registrations_ending_at_T_gt": ok}
            del dev
        except Exception as e:  # never lose the headline line to an auxiliary leg
            out["mapbuild_1gpu"] = {"error": repr(e)}

    # ---- CPU baseline leg (N = 1 only): the oracle on the same inputs ----
    if world == 1 and not args.no_cpu_baseline and workload == "single":
        from oracle import pyoracle as po
        hi_ = host_info()
        out["host"] = hi_
        phys = max(1, hi_["physical_cores"])
        granted = hi_["granted_cpus"]
        # THE baseline: the faithful port with as many threads as the box really runs -- min(physical cores, floor(cgroup CPU
        # quota)).  Threads beyond the quota are throttled by the kernel, not run, and make the port SLOWER.
        med, n_t, tb, r = timed_oracle(po, tgt, src, RESOLUTION, granted, False, 12.0)
        out["cpu_baseline"] = {"value": 1.0 / med, "unit": "registrations/s", "cores": granted, "kind": "port",
                               "sample": "%d full registrations of the same workload (median), after 1 warm-up; align only, target grid resident" % n_t,
                               "ms_per_registration": med * 1e3, "target_build_ms": tb * 1e3, "evaluations": r["n_evals"],
                               "cpu_model": hi_["cpu_model"], "threads": granted, "host_logical_cpus": hi_["logical_cpus"],
                               "host_physical_cores": phys, "cgroup_cpu_quota": hi_["cgroup_cpu_quota"],
                               "note": "threads = min(physical cores, floor(cgroup CPU quota)): what this box grants the process"}
        out["parity_vs_oracle"] = {"rot_max_abs": float(np.abs(T_timed[:3, :3] - r["T"][:3, :3]).max()),
                                   "trans_max_abs_m": float(np.abs(T_timed[:3, 3] - r["T"][:3, 3]).max()),
                                   "iterations_gpu": it_timed, "iterations_oracle": r["iterations"],
                                   "evals_gpu": st["n_evals"], "evals_oracle": r["n_evals"]}
        best_cpu = 1.0 / med
        if phys > granted:  # for the record: every physical core of the host, most of them throttled
            medp, np_, _, _ = timed_oracle(po, tgt, src, RESOLUTION, phys, False, 6.0)
            best_cpu = max(best_cpu, 1.0 / medp)  # (the ratio below is always against the fastest CPU variant measured)
            out["cpu_baseline_all_cores_throttled"] = {"value": 1.0 / medp, "unit": "registrations/s", "cores": phys, "threads": phys, "kind": "port",
                                                       "sample": "%d full registrations (median)" % np_, "ms_per_registration": medp * 1e3,
                                                       "note": "%d threads inside a cgroup that grants %s CPUs" % (phys, hi_["cgroup_cpu_quota"])}
        # also an OPTIMISED CPU variant, so that the ratio is not inflated by the reference's own
        # inefficiencies (rb-tree voxel lookup, N x 344 B scratch allocated, zeroed and summed per evaluation,
        # serial f64 Hessian).  Same arithmetic per neighbour.
        opt = {}
        for thr in sorted({granted, min(16, hi_["logical_cpus"])}):
            med_o, n_o, _, ro = timed_oracle(po, tgt, src, RESOLUTION, thr, True, 6.0)
            opt[thr] = {"value": 1.0 / med_o, "ms_per_registration": med_o * 1e3, "threads": thr, "runs": n_o, "evaluations": ro["n_evals"]}
            best_cpu = max(best_cpu, 1.0 / med_o)
        best_thr = max(opt, key=lambda t: opt[t]["value"])
        out["cpu_baseline_optimised"] = {
            "value": opt[best_thr]["value"], "unit": "registrations/s", "cores": best_thr, "threads": best_thr, "kind": "port-optimised",
            "sample": "%d full registrations (median) of the same workload, best of the thread counts tried" % opt[best_thr]["runs"],
            "ms_per_registration": opt[best_thr]["ms_per_registration"], "by_threads": opt, "cpu_model": hi_["cpu_model"],
            "what": "dense voxel lookup, per-thread accumulators instead of per-point result arrays, point "
                    "derivatives once per point, parallel f64 Hessian"}
        out["speedup_vs_cpu_baseline"] = value / out["cpu_baseline"]["value"]
        out["speedup_vs_best_cpu_variant"] = value / best_cpu


# =====================================================================================================
# legs of the batch workloads: roofline of the batch kernels (rank 0) and the RCCL lock-step form (all ranks)
# =====================================================================================================
def batch_legs(out, args, reg, nd, ndt, dist, rank, world, workload, step, steps, dt, N_SOURCE, dev, offsets, lo, hi, T_gts, torch):
    import numpy as np
    n_local = hi - lo
    if rank == 0 and n_local:
        # one pass of the same step with a HIP event pair on the library's stream around the derivative kernels of every
        # lock-step (ndt_profile_enable(1)); the algorithmic bytes are 8(d)'s per-evaluation figure x the scan evaluations served
        reg.profile(1)
        reg.profile_read(0)
        step()
        n_steps_timed, ms = reg.profile_read(0)
        reg.profile(0)
        stb = reg.stats()
        n_scan_evals = stb["n_evals"] + stb["n_hessian_recomputes"]
        bytes_total = n_scan_evals * algorithmic_bytes_per_eval(N_SOURCE, stb["mean_neighbors"])
        achieved = bytes_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        out["scan_evaluations_per_step_rank0"] = n_scan_evals
        out["mean_neighbors"] = stb["mean_neighbors"]
        committed = {}
        prof, src_name = committed_profile("batch64_summary.json")
        if prof and prof.get("derivative_kernels_hbm_bytes_per_scan_evaluation") and args.set == "U":
            per_eval = prof["derivative_kernels_hbm_bytes_per_scan_evaluation"]
            committed = {"traffic_committed_profile": per_eval * n_scan_evals / max(n_steps_timed, 1),
                         "traffic_committed_profile_per_scan_evaluation": per_eval,
                         "traffic_committed_profile_source": src_name + " (tools/profile_batch.sh + tools/summarize_batch_profile.py: separate --pmc "
                                                                        "FETCH_SIZE / WRITE_SIZE passes of the 64-scan command, KB, 2 x FETCH_SIZE on gfx950; per scan "
                                                                        "evaluation x the scan evaluations of this run's launches)"}
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                           "algorithmic_bytes_per_launch": bytes_total / max(n_steps_timed, 1),
                           "kernel": "k_derivatives<DIRECT7> / k_batch_step (one launch per lock-step over every live scan of rank 0's share; "
                                     "+ k_hessian64 where a scan's line search iterated)",
                           "avg_kernel_us": ms * 1e3 / max(n_steps_timed, 1), "launches_timed": n_steps_timed,
                           "scan_evaluations": n_scan_evals, "scans_on_rank0": n_local,
                           "algorithmic_bytes_per_scan_evaluation": algorithmic_bytes_per_eval(N_SOURCE, stb["mean_neighbors"]),
                           "kernel_time_share_of_step": ms / (dt / steps * 1e3),
                           "how": "one hipEvent pair per lock-step on the library stream around the derivative kernels "
                                  "(ndt_profile_enable(1)), one extra pass of the same step"}
        out["roofline"].update(committed)
    # ---- the literal north_star form: scans sharded, ONE RCCL all-reduce of the [n_scans][32] f64 rows per lock-step ----
    if workload == "mapbuild" and world > 1 and not args.no_lockstep_leg:
        result = {}
        done = threading.Event()

        def watchdog():  # a collective that never completes must not take the measured line down with it
            if not done.wait(float(os.environ.get("NDT_BENCH_LOCKSTEP_BUDGET_S", "240"))):
                if rank == 0:
                    out["lockstep_allreduce"] = {"error": "timed out: a collective of the lock-step leg never returned"}
                    print(json.dumps(out), flush=True)
                os._exit(3)  # the measured line is out; the run still FAILED (rc != 0)
        threading.Thread(target=watchdog, daemon=True).start()
        try:
            on_dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
            uid = torch.zeros(ndt.COMM_ID_BYTES, dtype=torch.uint8)
            if rank == 0:
                uid = torch.from_numpy(np.frombuffer(ndt.comm_get_unique_id(), dtype=np.uint8).copy())
            uid = uid.to(on_dev)
            dist.broadcast(uid, 0)
            def all_ranks_ok(step, fn):
                """Runs fn on this rank; every rank then learns whether ALL ranks got through (one torch all-reduce, which every
                rank reaches whatever fn did): a rank that failed must not leave the others waiting in the next collective."""
                err = None
                try:
                    fn()
                except Exception as e:  # noqa: BLE001 -- reported below, on every rank
                    err = repr(e)
                flag = torch.tensor([0.0 if err is None else 1.0], dtype=torch.float64, device=on_dev)
                dist.all_reduce(flag)
                if flag.item() != 0:
                    raise RuntimeError("%s failed on %d of %d ranks%s" % (step, int(flag.item()), world, "" if err is None else " (this rank: %s)" % err))

            # the transport of the one all-reduce per lock-step: RCCL from C++ on the library's stream (ndt_comm_*) -- or, in the
            # one-GPU rehearsal (RCCL refuses two ranks on one device), the same exchange through ndt_set_allreduce over gloo, so
            # that the rehearsal's line carries every key the real line does
            hook_calls = [0]
            if args.rehearse_one_gpu:
                transport = "ndt_set_allreduce hook over gloo (one-GPU rehearsal; the real run: RCCL, ndt_comm_*)"
                inner = nd.make_allreduce()

                def counted(addr, n, on_device):
                    hook_calls[0] += 1
                    return inner(addr, n, on_device)
                all_ranks_ok("ndt_set_allreduce", lambda: reg.setAllreduce(counted, on_device=False))
            else:
                transport = "RCCL (ncclAllReduce from C++ on the library's stream, ndt_comm_*)"
                all_ranks_ok("ndt_comm_init_rank", lambda: reg.commInitRank(bytes(uid.cpu().numpy().tobytes()), rank, world))

            def collectives_so_far():
                return hook_calls[0] if args.rehearse_one_gpu else reg.commStats()["collectives"]
            kw = dict(device_ptr=dev.data_ptr() if dev is not None else 0, offsets=offsets, stride_bytes=16,
                      first_scan=lo, total_scans=args.scans)
            warm = {}
            all_ranks_ok("the first sharded lock-step batch", lambda: warm.update(res=reg.alignBatchSharded(**kw)))  # warm-up (communicator set-up, first collective)
            res = warm["res"]
            torch.cuda.synchronize()
            n_ls = max(1, args.lockstep_steps)
            thr0 = cgroup_throttle_counters()
            dist.barrier()
            ta = time.perf_counter()
            for _ in range(n_ls):
                res = reg.alignBatchSharded(**kw)
            torch.cuda.synchronize()
            dist.barrier()
            tl = (time.perf_counter() - ta) / n_ls
            thr1 = cgroup_throttle_counters()
            t = torch.tensor([tl], dtype=torch.float64, device=on_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ok = 0
            for k, Tg in enumerate(T_gts):
                ok += int(near_T_gt(res["T"][lo + k], Tg))
            okt = torch.tensor([float(ok)], dtype=torch.float64, device=on_dev)
            dist.all_reduce(okt)
            cs = reg.commStats()
            # one more pass with HIP events on the library stream: per lock-step, the derivative kernels / k_reduce +
            # ncclAllReduce / k_publish_rows, and the host's wall time for the whole step (ndt_profile_enable(1), kinds 0, 4, 5, 6)
            reg.profile(1)
            for k in (0, 4, 5, 6):
                reg.profile_read(k)
            c_before = collectives_so_far()
            reg.alignBatchSharded(**kw)
            torch.cuda.synchronize()
            c_batch = collectives_so_far() - c_before
            cs = reg.commStats()
            split = {k: reg.profile_read(k) for k in (0, 4, 5, 6)}
            reg.profile(0)
            n_st = max(split[6][0], 1)
            mine = torch.zeros((world, 5), dtype=torch.float64, device=on_dev)
            mine[rank, 0] = split[0][1] / n_st * 1e3
            mine[rank, 1] = split[4][1] / max(split[4][0], 1) * 1e3
            mine[rank, 2] = split[5][1] / max(split[5][0], 1) * 1e3
            mine[rank, 3] = split[6][1] / n_st * 1e3
            mine[rank, 4] = float((thr1[0] - thr0[0]) if (thr0 and thr1) else -1)
            dist.all_reduce(mine)
            per_rank = mine.cpu().numpy()
            result = {"value": args.scans / float(t.item()), "unit": "registrations/s", "ms_per_step": float(t.item()) * 1e3,
                      "steps": n_ls, "registrations_ending_at_T_gt": int(okt.item()), "lock_steps": cs["lock_steps"],
                      "collectives": int(c_batch), "collectives_equal_lock_steps_plus_1": bool(c_batch == cs["lock_steps"] + 1),
                      "transport": transport,
                      "allreduce_doubles_per_lock_step": args.scans * 32,
                      "rccl_world_size": (cs["world"] if not args.rehearse_one_gpu else None), "world_size": world,
                      "per_lock_step_us": {"lock_steps_profiled": int(n_st),
                                           "derivative_kernels": [float(x) for x in per_rank[:, 0]],
                                           "reduce_plus_ncclAllReduce": [float(x) for x in per_rank[:, 1]],
                                           "publish_rows": [float(x) for x in per_rank[:, 2]],
                                           "host_wall": [float(x) for x in per_rank[:, 3]],
                                           "how": "per rank; HIP events on the library stream in one extra profiled pass (ndt_profile_enable(1): "
                                                  "kinds 0 / 4 / 5) and the host's wall clock per lock-step (kind 6)"},
                      "cgroup_nr_throttled_delta_per_rank": [int(x) for x in per_rank[:, 4]],
                      "host_thread_plan": dict(zip(("affinity_cpus", "cgroup_quota_cpus", "local_world_size"), ndt.host_thread_budget())),
                      "what": "every rank steps all %d solvers; rows of the scans a rank does not own are zero; one in-place "
                              "ncclAllReduce(sum) of the [%d][32] f64 buffer per lock-step on the library stream (C++, ndt_comm_*)" % (args.scans, args.scans)}
            result["host_thread_plan"]["pool_threads"], result["host_thread_plan"]["max_batch_groups"] = ndt.host_thread_plan(*ndt.host_thread_budget())
            if args.rehearse_one_gpu:
                reg.setAllreduce(None)
            else:
                reg.commDestroy()
        except Exception as e:
            result = {"error": repr(e)}
        done.set()
        if rank == 0:
            out["lockstep_allreduce"] = result


# =====================================================================================================
# configs[4]: multi-resolution NDT (2.0 -> 1.0 -> 0.5 m) on a streamed sequence of 2M-pt PCD scans
# =====================================================================================================
def run_pyramid(args, ndt, clouds, tgt, device, steps, warmup, binding, world=1, rank=0, barrier=lambda: None, max_over_ranks=lambda x: x):
    """configs[4]: every rank streams a sequence of its own (its own directory of PCD files) through three resident grids --
    independent sequences, no collective (weak scaling).  A step is one pass over the rank's whole sequence."""
    import shutil
    import tempfile
    import numpy as np
    from toyslam_amd import pyramid
    n_scans, n_src = args.seq_scans, 2000000
    tmp = tempfile.mkdtemp(prefix="ndt_seq_r%d_" % rank)
    T_gts = pyramid.write_sequence(tmp, tgt, n_scans, n_src, seed=clouds.SEED + 2000 + 100000 * rank)
    pyr = pyramid.Pyramid(levels=(2.0, 1.0, 0.5), device=device)
    t0 = time.perf_counter()
    pyr.setInputTarget(tgt)
    t_build = time.perf_counter() - t0
    steps = max(1, steps)
    for _ in range(max(1, warmup)):  # (the first pass also fills the page cache and grows the page-locked buffers)
        r = pyr.run_sequence(tmp)
    barrier()
    t0 = time.perf_counter()
    pass_ms = []
    for _ in range(steps):
        t1 = time.perf_counter()
        r = pyr.run_sequence(tmp)
        pass_ms.append((time.perf_counter() - t1) * 1e3)
    barrier()
    dt = max_over_ranks(time.perf_counter() - t0)
    shutil.rmtree(tmp, ignore_errors=True)
    ok = sum(int(np.abs(T[:3, :3] - Tg[:3, :3]).max() < 5e-4 and np.abs(T[:3, 3] - Tg[:3, 3]).max() < 2e-2) for T, Tg in zip(r["T"], T_gts))
    return {"metric": "scans/sec (2M-pt scans, 2.0->1.0->0.5 m pyramid vs 10M-pt target, streamed PCD sequence)",
            "value": world * steps * n_scans / dt, "unit": "scans/s", "n_gpus": world, "steps": steps, "warmup": max(1, warmup),
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "configs[4]: %d x 2M-pt PCD scans per GPU streamed from disk (five files read and parsed ahead, page-locked "
                                   "buffers), three resident grids 2.0 / 1.0 / 0.5 m over one 10M-pt target (%g m scene), each level's result "
                                   "the next level's guess; one independent sequence per rank" % (n_scans, args.extent)},
            "world_size": world, "pass_ms_rank0": pass_ms, "per_level_ms": r["per_level_ms"], "upload_ms_per_scan": r["upload_ms"], "wait_for_file_ms_per_scan": r["wait_ms"],
            "grids_build_ms": t_build * 1e3, "scans_ending_at_T_gt_rank0": ok, "evaluations_per_scan": r["evals_per_scan"],
            "host_binding": binding}


if __name__ == "__main__":
    main()
