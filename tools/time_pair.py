"""The mapping nodes' real size: the reference scan pair (16k points each), per-call wall times of the calls a node makes per
scan -- setInputTarget, setInputSource, align (with and without fetching the aligned cloud), getFitnessScore (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()  # (before the library touches the device: initialised later, torch finds no GPU)
from toyslam_amd import ndt
d = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "pair_0p1.npz"))
t, s = d["target"], d["source"]
g = ndt.NormalDistributionsTransform(); g.setResolution(1.0)
g.setInputTarget(t); g.setInputSource(s); g.align()
def med(f, n=30):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append((time.perf_counter() - t0) * 1e6)
    return float(np.median(ts))
print("setInputTarget %.0f us | setInputSource %.0f us" % (med(lambda: g.setInputTarget(t)), med(lambda: g.setInputSource(s))))
print("align (no cloud) %.0f us | align + aligned cloud to host %.0f us | iterations %d evals %d" %
      (med(lambda: g.align()), med(lambda: g.align(n_out=len(s))), g.getFinalNumIteration(), g.stats()["n_evals"]))
print("getFitnessScore %.0f us" % med(lambda: g.getFitnessScore()))
def scan():
    g.setInputTarget(t); g.setInputSource(s); g.align(); g.getFinalTransformation()
per_scan = med(scan)
print("per scan (target + source + align + result) %.0f us" % per_scan)
# the host's part of the two set calls with the GPU idle when they start, and call + the GPU's completion (the back-to-back
# figures below are paced by the GPU: a call returns once its work is queued and the page-locked slot of four calls ago is free)
def split(f, n=60):
    a, b = [], []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter(); f(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        a.append((t1 - t0) * 1e6); b.append((t2 - t0) * 1e6)
    return float(np.median(a)), float(np.median(b))
tc, tg = split(lambda: g.setInputTarget(t))
sc, sg = split(lambda: g.setInputSource(s))
def scan_parts():
    t0 = time.perf_counter(); g.setInputTarget(t); t1 = time.perf_counter(); g.setInputSource(s); t2 = time.perf_counter(); g.align(); t3 = time.perf_counter()
    g.getFinalTransformation(); t4 = time.perf_counter()
    return [(t1 - t0) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6]
parts = [float(x) for x in np.median(np.array([scan_parts() for _ in range(60)]), axis=0)]
import json
print(json.dumps({"set_input_target_call_us": tc, "set_input_target_call_and_gpu_us": tg, "set_input_source_call_us": sc,
                  "set_input_source_call_and_gpu_us": sg, "in_loop_target_source_align_result_us": parts, "workload": "reference pair after the 0.1 m prefilter (15772 / 15950 points), resolution 1.0, DIRECT7, class defaults",
                  "set_input_target_us": med(lambda: g.setInputTarget(t)), "set_input_source_us": med(lambda: g.setInputSource(s)),
                  "align_us": med(lambda: g.align()), "align_with_cloud_to_host_us": med(lambda: g.align(n_out=len(s))),
                  "get_fitness_score_us": med(lambda: g.getFitnessScore()), "per_scan_us": per_scan,
                  "iterations": g.getFinalNumIteration(), "evaluations": g.stats()["n_evals"]}))
