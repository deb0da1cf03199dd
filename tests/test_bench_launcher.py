"""bench.py --gpus N must run N ranks whichever way it is started (`python bench.py
--gpus 8` used to run one rank and report n_gpus = 1).  --dry swaps the workload for one gloo
all-reduce on the CPU, so the launcher is testable without GPUs."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_line(stdout):
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_plain_start_spawns_the_ranks_itself():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["ranks_seen"] == 2 and line["backend"] == "gloo"


def test_one_gpu_start_is_unchanged_and_spawns_nothing():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, BENCH, "--dry"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _json_line(r.stdout)
    assert line["n_gpus"] == 1 and line["ranks_seen"] == 1 and line["backend"] is None


def test_under_torchrun_the_flag_must_agree_with_the_world_size():
    base = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
            "--master-addr", "127.0.0.1"]
    ok = subprocess.run(base + ["--master-port", "29541", BENCH, "--gpus", "2", "--dry"], capture_output=True,
                        text=True, timeout=300)
    assert ok.returncode == 0, ok.stderr
    assert _json_line(ok.stdout)["ranks_seen"] == 2
    bad = subprocess.run(base + ["--master-port", "29542", BENCH, "--gpus", "4", "--dry"], capture_output=True,
                         text=True, timeout=300)
    assert bad.returncode != 0 and "WORLD_SIZE=2" in bad.stderr


def test_a_dead_rank_takes_the_launch_down_within_seconds():
    """VERDICT r2 / ADVICE: the parent used to block on rank 0 while rank 0 waited for a peer that had
    died.  Rank 1 exits before the rendezvous; rank 0 would wait for it until the process-group
    timeout: the watchdog must end the run non-zero long before that."""
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["DGNN_BENCH_DRY_FAIL_RANK"] = "1"
    env["DGNN_PG_TIMEOUT"] = "600"
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0
    assert "(1, 3)" in r.stderr, r.stderr
    assert time.monotonic() - t0 < 120


def test_the_launch_has_an_overall_time_limit():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    env["DGNN_BENCH_LAUNCH_TIMEOUT"] = "0.01"
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry"], capture_output=True, text=True,
                       env=env, timeout=300)
    assert r.returncode != 0 and "no result after" in r.stderr
