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
