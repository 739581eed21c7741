"""bench.py's N > 1 launcher on CPU: the per-rank environment plan, the self-launch of N child ranks (no exec,
before anything touches a GPU) relaying rank 0's one JSON line, and the same entry point under
torch.distributed.run -- with the GPU-free `selftest` workload over gloo."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(cmd, env=None, timeout=300):
    e = dict(os.environ)
    e.pop("WORLD_SIZE", None)
    e.pop("RANK", None)
    e.pop("LOCAL_RANK", None)
    if env:
        e.update(env)
    return subprocess.run(cmd, env=e, capture_output=True, text=True, timeout=timeout)


def test_rank_environment_plan():
    sys.path.insert(0, ROOT)
    import bench
    envs = bench.rank_environments(4, 29511, base_env={"PATH": "/usr/bin"})
    assert [e["RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert [e["LOCAL_RANK"] for e in envs] == ["0", "1", "2", "3"]
    assert all(e["WORLD_SIZE"] == "4" and e["MASTER_ADDR"] == "127.0.0.1" and e["MASTER_PORT"] == "29511" for e in envs)
    assert all(e["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" for e in envs)  # dmabuf IPC only on this pool


def test_dry_run_prints_the_plan_and_launches_nothing():
    r = _run([sys.executable, BENCH, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"])
    assert r.returncode == 0, r.stderr
    plan = json.loads(r.stdout.strip().splitlines()[-1])
    assert plan["n_ranks"] == 8 and len(plan["ranks"]) == 8
    assert plan["command"][1].endswith("bench.py") and "--gpus" in plan["command"]


def test_default_workload_by_rank_count():
    sys.path.insert(0, ROOT)
    import bench
    a = bench.parse_args(["--gpus", "8"])
    assert a.workload == "auto" and a.scans == 512  # auto -> single at N = 1, mapbuild (configs[3]: 512 scans) at N > 1


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_relays_rank0_line(n):
    r = _run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--warmup", "0", "--workload", "selftest", "--scans", "512"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1  # ONE JSON line, rank 0's
    out = json.loads(lines[0])
    assert out["world_size"] == n and out["backend"] == "gloo"
    assert out["scans_covered"] == 512  # shard_range covers the batch exactly once across the ranks


def test_same_entry_point_under_torch_distributed_run():
    """The form the driver uses for N > 1."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(port), BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--workload", "selftest"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["world_size"] == 2


def test_self_launch_stops_the_other_ranks_when_one_dies(tmp_path):
    """A rank that dies before the rendezvous (a HIP init failure on one GPU) must not leave the launcher waiting for ever
    with the other ranks alive: bench.supervise terminates the survivors and reports failure."""
    sys.path.insert(0, ROOT)
    import time
    import bench
    sleeper = [sys.executable, "-c", "import time; time.sleep(600)"]
    procs = [subprocess.Popen(sleeper), subprocess.Popen([sys.executable, "-c", "import sys; sys.exit(3)"]), subprocess.Popen(sleeper)]
    t0 = time.monotonic()
    try:
        rcs = bench.supervise(procs, budget_s=120.0, grace_s=5.0, poll_s=0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert time.monotonic() - t0 < 30
    assert all(p.poll() is not None for p in procs)
    assert rcs[1] == 3 and all(c != 0 for c in rcs)
    # ... and the wall-clock budget: healthy but endless ranks are stopped too
    procs = [subprocess.Popen(sleeper), subprocess.Popen(sleeper)]
    try:
        rcs = bench.supervise(procs, budget_s=1.0, grace_s=5.0, poll_s=0.05)
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    assert all(p.poll() is not None for p in procs) and all(c != 0 for c in rcs)
    # ... and the good case is left alone
    procs = [subprocess.Popen([sys.executable, "-c", "pass"]) for _ in range(3)]
    assert bench.supervise(procs, budget_s=60.0) == [0, 0, 0]


def test_a_rank_lost_after_the_rendezvous_fails_the_launch_within_its_budget():
    """A rank that dies in the MIDDLE of a run -- after the rendezvous and a first collective, the state a rank is in after
    ndt_comm_init_rank -- leaves the others inside the next collective: the self-launcher must notice the exit code, stop the
    survivors and return non-zero, well within its budget; no JSON line claims a result."""
    import time
    t0 = time.monotonic()
    r = _run([sys.executable, BENCH, "--gpus", "3", "--steps", "2", "--warmup", "0", "--workload", "selftest", "--selftest-die-rank", "1"],
             env={"NDT_BENCH_LAUNCH_BUDGET_S": "120"}, timeout=200)
    assert r.returncode != 0
    assert time.monotonic() - t0 < 90
    assert "stopping the remaining ranks" in r.stderr or "rank exit codes" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{") and "scans_covered" in ln]
