#!/usr/bin/env python3
"""bench.py -- scan registrations/sec of the MI355X NDT core (BASELINE.json metric).

A "step" is ONE registration (pcl::Registration::align) of one synthetic 100k-point source scan
against a 1M-point target whose voxel grid is already resident in HBM -- exactly the region
ndt_omp/apps/align.cpp:20-29 times -- at 1.0 m voxels, DIRECT7, with the Newton loop pinned to
30 outer passes (max_iterations 28, transformation_epsilon 0: ndt_omp_impl.hpp:158-164 then runs
max_iterations + 2 passes).  configs[1] of BASELINE.json.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--set U|S] [--workload single|batch]

N > 1 (launched by torch.distributed.run, one rank per GPU): registrations of different scans are
independent, so every rank registers its own scan against its own replica of the target grid with
no collective in the data path ("scaling": "weak"); value = N*K / max-over-ranks time.

Rank 0 prints ONE JSON line.  Extra legs on rank 0 (outside the timed region):
  roofline     -- average duration of the dominant kernel (k_eval_server: one persistent launch per
                  registration) from HIP events on the library's own stream, over a second pass of
                  the same steps; algorithmic bytes = evaluations served x bytes per evaluation.
  cpu_baseline -- the oracle (oracle/, a faithful OpenMP port of the reference's algorithm; the real
                  pclomp cannot be built here) on the same inputs on the host cores (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

METRIC = "scan registrations/sec (100k-pt src vs 1M-pt target, 30 Newton iters)"
M_TARGET, N_SOURCE, RESOLUTION, MAX_ITER, EPS = 1000000, 100000, 1.0, 28, 0.0
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md)


def algorithmic_bytes_per_eval(n_points, mean_neighbors):
    """16 B point + 7 x 4 B voxel-slot probes + 36 B per valid neighbour record."""
    return n_points * (16 + 7 * 4 + 36 * mean_neighbors)


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--set", default="U", choices=["U", "S"], help="U = uniform box (headline), S = surface scene")
    ap.add_argument("--workload", default="single", choices=["single", "batch", "large"],
                    help="single = configs[1] (headline); batch = configs[3] shape per GPU; "
                         "large = configs[2]: 2M-pt source vs 10M-pt target, 0.5 m voxels")
    ap.add_argument("--batch", type=int, default=64, help="scans per GPU for --workload batch (config 4 shape)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif torch.cuda.is_available():
        torch.cuda.set_device(local_rank)

    binding = bind_near_gpu(local_rank) if torch.cuda.is_available() else None

    from toyslam_amd import clouds, ndt

    def barrier():
        if torch.cuda.is_available():
            torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # ---- inputs (synthetic, seeded; resident in HBM before the timed region) ----
    global M_TARGET, N_SOURCE, RESOLUTION
    if args.workload == "large":
        M_TARGET, N_SOURCE, RESOLUTION = 10000000, 2000000, 0.5
    if args.workload == "large":   # configs[2]: surface scene of 200 x 200 m (at 400 m a 10M-pt map is too sparse for 0.5 m voxels: most hold < 6 points)
        tgt = clouds.target_surfaces(M_TARGET, extent=200.0, n_boxes=120)
    else:
        tgt = clouds.target_uniform(M_TARGET) if args.set == "U" else clouds.target_surfaces(M_TARGET)
    reg = ndt.NormalDistributionsTransform(device=local_rank)
    reg.setResolution(RESOLUTION)
    reg.setNeighborhoodSearchMethod(ndt.DIRECT7)
    reg.setMaximumIterations(MAX_ITER)
    reg.setTransformationEpsilon(EPS)
    t0 = time.perf_counter()
    reg.setInputTarget(tgt)
    t_build_first = time.perf_counter() - t0

    def best_of(fn, n=5):
        ts = []
        for _ in range(n):
            ta = time.perf_counter()
            fn()
            ts.append(time.perf_counter() - ta)
        return float(np.median(ts))
    t_build = best_of(lambda: reg.setInputTarget(tgt))                 # host buffer: H2D over PCIe + grid build
    tgt_dev = torch.from_numpy(np.c_[tgt, np.ones(len(tgt), np.float32)]).cuda()
    torch.cuda.synchronize()
    t_build_dev = best_of(lambda: reg.setInputTargetDevice(tgt_dev.data_ptr(), len(tgt), 16))  # cloud already in HBM

    if args.workload in ("single", "large"):
        src = clouds.source_from_target(tgt, N_SOURCE, seed=clouds.SEED + 1 + 7 * rank)
        reg.setInputSource(src)
        t_source = best_of(lambda: reg.setInputSource(src))           # H2D + spatial ordering of the scan

        def step():
            reg.align()
        regs_per_step = 1
    else:
        rng = np.random.default_rng(clouds.SEED + 100 + rank)
        scans = []
        for k in range(args.batch):
            T = clouds.random_T(rng, 0.5, 2.0)
            scans.append(clouds.source_from_target(tgt, N_SOURCE, T_gt=T, seed=clouds.SEED + 1000 * rank + k))
        cat = np.ascontiguousarray(np.concatenate(scans, axis=0))
        dev = torch.from_numpy(np.c_[cat, np.ones(len(cat), np.float32)]).cuda()
        offsets = np.arange(args.batch + 1, dtype=np.uintp) * N_SOURCE

        def step():
            reg.alignBatch(device_ptr=dev.data_ptr(), offsets=offsets, stride_bytes=16)
        regs_per_step = args.batch

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        st = reg.stats() if args.workload in ("single", "large") else {}
        T_timed = reg.getFinalTransformation() if args.workload in ("single", "large") else None
        it_timed = reg.getFinalNumIteration() if args.workload in ("single", "large") else None
        value = world * args.steps * regs_per_step / dt
        out = {
            "metric": METRIC, "value": value, "unit": "registrations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("single 100k-pt source vs 1M-pt target, 1.0 m voxels, DIRECT7, 30 Newton passes "
                                    "(max_iterations 28, epsilon 0), set " + args.set) if args.workload == "single" else
                       ("single 2M-pt source vs 10M-pt target (surface scene, 200 m), 0.5 m voxels, DIRECT7, 30 Newton passes")
                       if args.workload == "large" else
                       ("map-build batch: %d x 100k-pt sources per GPU vs one 1M-pt target, lock-step, set %s" % (args.batch, args.set)),
                       "target_points": M_TARGET, "source_points": N_SOURCE, "resolution_m": RESOLUTION,
                       "search": "DIRECT7", "outer_passes": MAX_ITER + 2, "sharding": "one scan stream per GPU, target grid replicated"},
            "host_binding": binding, "target_build_ms": t_build * 1e3,
            "target_build_roofline": None, "target_build_device_resident_ms": t_build_dev * 1e3,
            "target_build_first_call_ms": t_build_first * 1e3,
        }
        try:  # K1 against its own roof (M x 16 B read + V x 64 B of records written)
            gi = reg.grid_counts()
            k1_bytes = M_TARGET * 16 + gi["n_leaves"] * 64
            out["target_build_roofline"] = {"bound": "hbm", "algorithmic_bytes": k1_bytes,
                                            "achieved": k1_bytes / t_build_dev / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                            "frac": k1_bytes / t_build_dev / 1e9 / HBM_PEAK_GBS,
                                            "occupied_voxels": gi["n_leaves"], "valid_voxels": gi["n_valid"],
                                            "note": "wall time of ndt_set_input_target_device: a chain of ~10 dependent kernels, not one "
                                                    "kernel; at 1M points their durations add up to the wall time "
                                                    "(profiles/r01_kernel_stats.csv: k_count 46 us of device-scope atomics, "
                                                    "k_finalize 38 us of per-voxel f64 eigen-decompositions, scatter / presort / "
                                                    "repack ~16 us each, scans ~20 us), at the nodes' 16k points launch latency"}
        except Exception:
            pass
        if args.workload in ("single", "large"):
            out["set_source_ms"] = t_source * 1e3
            out["registrations_per_s_incl_target_build_and_source_upload"] = 1.0 / (dt / args.steps + t_build + t_source)
            out["evaluations_per_registration"] = st["n_evals"]
            out["f64_hessian_recomputes"] = st["n_hessian_recomputes"]
            out["mean_neighbors"] = st["mean_neighbors"]
            # ---- roofline leg: HIP events on the library's stream, second pass of the same steps ----
            # The kernel of the timed region is k_eval_server: ONE launch per registration that serves
            # every evaluation of it.  ndt_profile_enable(2) brackets that launch with an event pair.
            n_rep = min(args.steps, 10)
            reg.profile(2)
            reg.profile_read(3)
            for _ in range(n_rep):
                reg.align()
            n_launch, ms = reg.profile_read(3)
            st2 = reg.stats()
            # ... and, for reference, the same device code as one launch per evaluation (profile mode 1)
            reg.profile(1)
            reg.profile_read(0)
            for _ in range(n_rep):
                reg.align()
            n_eval_launch, ms_eval = reg.profile_read(0)
            reg.profile(0)
            hbar = st2["mean_neighbors"]
            avg_s = ms * 1e-3 / max(n_launch, 1)
            bytes_per_eval = algorithmic_bytes_per_eval(N_SOURCE, hbar)
            evals_per_launch = st2["n_evals"] + st2["n_hessian_recomputes"]
            bytes_per_launch = evals_per_launch * bytes_per_eval
            achieved = bytes_per_launch / avg_s / 1e9
            # HBM traffic of the same kernel from the committed PMC passes of this command
            # (tools/profile_round.sh: separate --pmc FETCH_SIZE / WRITE_SIZE runs; unit KB; gfx950
            # correction 2 x FETCH_SIZE, MI355X_MICROARCH.md "HBM") -- not collectable live here.
            traffic, traffic_src = None, None
            try:
                prof = json.load(open(os.path.join(ROOT, "profiles", "r01_profile_summary.json")))
                for name, c in prof["pmc"].items():
                    if "k_eval_server<7>" in name and args.workload == "single":
                        traffic = (2.0 * c["FETCH_SIZE"]["mean"] + c["WRITE_SIZE"]["mean"]) * 1024.0
                        traffic_src = "profiles/r01_profile_summary.json"
            except Exception:
                pass
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "kernel": "k_eval_server<DIRECT7> (persistent: all evaluations of one registration, "
                                         "their reductions, the f64 Hessian recompute and the output transform)",
                               "avg_kernel_us": avg_s * 1e6, "launches_timed": n_launch,
                               "evaluations_per_launch": evals_per_launch,
                               "algorithmic_bytes_per_evaluation": bytes_per_eval,
                               "algorithmic_bytes_per_launch": bytes_per_launch, "mean_neighbors": hbar,
                               "how": "one hipEvent pair on the library stream around the kernel of each registration "
                                      "(ndt_profile_enable(2)), in a second pass of the same steps",
                               "per_evaluation_kernel": {
                                   "kernel": "k_derivatives_fused<DIRECT7, hessian>: the same device code as one launch "
                                             "per evaluation (ndt_profile_enable(1))",
                                   "avg_event_us": ms_eval * 1e3 / max(n_eval_launch, 1), "launches_timed": n_eval_launch,
                                   "achieved_GBs": bytes_per_eval / (ms_eval * 1e-3 / max(n_eval_launch, 1)) / 1e9}}
            out["us_per_evaluation_in_timed_region"] = dt / args.steps / max(st["n_evals"] + st["n_hessian_recomputes"], 1) * 1e6
            out["registration_algorithmic_GBs"] = (st["n_evals"] + st["n_hessian_recomputes"]) * bytes_per_eval / (dt / args.steps) / 1e9
            # ---- CPU baseline leg (N = 1 only): the oracle on the same inputs ----
            if world == 1 and not args.no_cpu_baseline and args.workload == "single":
                from oracle import pyoracle as po
                # the GPU box exposes every host core, but one GPU's share of it is 16 (and the port
                # does not scale past that: its zero-fill/ordered-reduce/f64-Hessian parts are serial)
                cores = min(16, len(os.sched_getaffinity(0)))
                o = po.OracleNDT(resolution=RESOLUTION, search_method=po.DIRECT7, num_threads=cores,
                                 trans_eps=EPS, max_iter=MAX_ITER)
                tb = time.perf_counter()
                o.set_target(tgt)
                tb = time.perf_counter() - tb
                o.set_source(src)
                r = o.align()  # warm-up, also the parity reference
                times = []
                budget = time.perf_counter() + 20.0
                while len(times) < 5 and (not times or time.perf_counter() < budget):
                    ta = time.perf_counter()
                    o.align()
                    times.append(time.perf_counter() - ta)
                med = float(np.median(times))
                T = T_timed  # result of the timed region's last registration
                out["cpu_baseline"] = {"value": 1.0 / med, "unit": "registrations/s", "cores": cores, "kind": "port",
                                       "sample": "%d full registrations of the same workload (median), after 1 warm-up; "
                                                 "align only, target grid resident" % len(times),
                                       "ms_per_registration": med * 1e3, "target_build_ms": tb * 1e3,
                                       "evaluations": r["n_evals"], "host_cores_visible": len(os.sched_getaffinity(0))}
                out["parity_vs_oracle"] = {"rot_max_abs": float(np.abs(T[:3, :3] - r["T"][:3, :3]).max()),
                                           "trans_max_abs_m": float(np.abs(T[:3, 3] - r["T"][:3, 3]).max()),
                                           "iterations_gpu": it_timed, "iterations_oracle": r["iterations"],
                                           "evals_gpu": st["n_evals"], "evals_oracle": r["n_evals"]}
                out["speedup_vs_cpu_baseline"] = value / (1.0 / med)
                # also an OPTIMISED CPU variant, so that the ratio is not inflated by the
                # reference's own inefficiencies (rb-tree voxel lookup, N x 344 B scratch allocated, zeroed
                # and summed per evaluation, serial f64 Hessian).  Same arithmetic per neighbour.
                oo = po.OracleNDT(resolution=RESOLUTION, search_method=po.DIRECT7, num_threads=cores,
                                  trans_eps=EPS, max_iter=MAX_ITER, optimised=True)
                oo.set_target(tgt)
                oo.set_source(src)
                ro = oo.align()
                times = []
                budget = time.perf_counter() + 10.0
                while len(times) < 5 and (not times or time.perf_counter() < budget):
                    ta = time.perf_counter()
                    oo.align()
                    times.append(time.perf_counter() - ta)
                med_o = float(np.median(times))
                out["cpu_baseline_optimised"] = {
                    "value": 1.0 / med_o, "unit": "registrations/s", "cores": cores, "kind": "port-optimised",
                    "sample": "%d full registrations (median) of the same workload" % len(times),
                    "ms_per_registration": med_o * 1e3, "evaluations": ro["n_evals"],
                    "what": "dense voxel lookup, per-thread accumulators instead of per-point result arrays, point "
                            "derivatives once per point, parallel f64 Hessian",
                    "speedup_of_gpu": value * med_o}
        if args.workload == "batch":
            # ---- roofline leg of the lock-step batch: one pass of the same step with a HIP event pair on the library's
            # stream around the derivative kernels of every lock-step (ndt_profile_enable(1)); the algorithmic bytes are
            # 8(d)'s per-evaluation figure x the scan evaluations those kernels served ----
            reg.profile(1)
            reg.profile_read(0)
            step()
            n_steps_timed, ms = reg.profile_read(0)
            reg.profile(0)
            stb = reg.stats()
            n_scan_evals = stb["n_evals"] + stb["n_hessian_recomputes"]
            bytes_total = n_scan_evals * algorithmic_bytes_per_eval(N_SOURCE, stb["mean_neighbors"])
            achieved = bytes_total / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            out["scan_evaluations_per_step"] = n_scan_evals
            out["mean_neighbors"] = stb["mean_neighbors"]
            traffic, traffic_src = None, None   # HBM bytes per lock-step from the committed PMC passes of the 64-scan command
            try:
                if args.batch == 64 and args.set == "U":
                    traffic = json.load(open(os.path.join(ROOT, "profiles", "r01_batch64_summary.json")))["derivative_kernels_hbm_bytes_per_lock_step"]
                    traffic_src = "profiles/r01_batch64_summary.json (tools/profile_batch.sh)"
            except Exception:
                pass
            out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                               "algorithmic_bytes_per_launch": bytes_total / max(n_steps_timed, 1),
                               "kernel": "k_derivatives<DIRECT7> / k_batch_step (one launch per lock-step over every live scan; "
                                         "+ k_hessian64 where a scan's line search iterated)",
                               "avg_kernel_us": ms * 1e3 / max(n_steps_timed, 1), "launches_timed": n_steps_timed,
                               "scan_evaluations": n_scan_evals,
                               "algorithmic_bytes_per_scan_evaluation": algorithmic_bytes_per_eval(N_SOURCE, stb["mean_neighbors"]),
                               "kernel_time_share_of_step": ms / (dt / args.steps * 1e3),
                               "how": "one hipEvent pair per lock-step on the library stream around the derivative kernels "
                                      "(ndt_profile_enable(1)), one extra pass of the same step"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
