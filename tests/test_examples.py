"""The example models (examples/: GAT Net, MoNet, DGCNN with the reference scripts' protocol and
CSV line) run end to end on the GPU, train (loss goes down on a fixed synthetic batch) and append
one CSV line in the reference's column layout."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run(script, *args, tmp_path):
    out = tmp_path / "result.csv"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", script), *args, "--output", str(out)],
                       capture_output=True, text=True, timeout=280)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    lines = out.read_text().strip().splitlines()
    assert len(lines) == 1
    loss = [ln for ln in p.stdout.splitlines() if ln.startswith("loss ")][0].split()
    return lines[0].split(","), float(loss[1]), float(loss[3].rstrip(";"))


def test_gat_example(tmp_path):
    cols, first, last = run("train_gatconv.py", "--dataset", "cora", "--n-heads", "4", "--n-hidden", "64",
                            "--n-epochs", "20", "--lr", "0.01", "--dropout", "0", tmp_path=tmp_path)
    assert cols[0] == "train_GAT_dgnn_amd" and cols[1] == "cora heads=4 hidden_dim=64" and len(cols) == 6
    assert cols[2].endswith("s") and cols[3].endswith("s") and cols[4].endswith("MB")
    assert last < first


def test_monet_example(tmp_path):
    cols, first, last = run("train_gmmconv.py", "--dataset", "cora", "--n-hidden", "64", "--n-epochs", "20",
                            "--dropout", "0", tmp_path=tmp_path)
    assert cols[0] == "train_GMM_dgnn_amd" and len(cols) == 5
    assert last < first


def test_dgcnn_example(tmp_path):
    cols, first, last = run("train_edgeconv.py", "--batch-size", "4", "--num-points", "256", "--k", "10",
                            "--n-epochs", "20", "--lr", "0.01", "--dropout", "0", tmp_path=tmp_path)
    assert cols[0] == "train_edgeconv_dgnn_amd" and cols[1] == "4 10" and len(cols) == 6
    assert last < first


def test_hip_graph_flag_refuses_dropout_before_touching_the_gpu(tmp_path):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "train_gatconv.py"), "--hip-graph"],
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "--dropout 0" in p.stderr


def test_gat_and_monet_examples_replayed_from_a_hip_graph(tmp_path):
    """--hip-graph: the training epoch captured once and replayed; it still trains.  Every run is a
    child process (`run`), so a capture gone wrong cannot take the test session's GPU context with it."""
    cols, first, last = run("train_gatconv.py", "--dataset", "cora", "--n-heads", "4", "--n-hidden", "64",
                            "--n-epochs", "30", "--lr", "0.01", "--dropout", "0", "--hip-graph", tmp_path=tmp_path)
    assert cols[1] == "cora heads=4 hidden_dim=64 hip-graph" and last < first
    (tmp_path / "result.csv").unlink()
    cols, first, last = run("train_gmmconv.py", "--dataset", "cora", "--n-hidden", "64", "--n-epochs", "30",
                            "--dropout", "0", "--hip-graph", tmp_path=tmp_path)
    assert cols[1].endswith("hip-graph") and last < first
    # attention dropout under capture: the operator's own generator (device {seed, counter} pair,
    # dgnn_gat_*_seeded_dev), a fresh mask per replay
    (tmp_path / "result.csv").unlink()
    cols, first, last = run("train_gatconv.py", "--dataset", "cora", "--n-heads", "4", "--n-hidden", "64",
                            "--n-epochs", "30", "--lr", "0.01", "--dropout", "0", "--attn-drop", "0.3", "--hip-graph",
                            tmp_path=tmp_path)
    assert cols[1] == "cora heads=4 hidden_dim=64 hip-graph" and last < first


def test_operator_dropout_captured_into_a_hip_graph_draws_a_fresh_mask_per_replay():
    """GATConvFuse with attn_drop = 0.5 captured after an eager warm-up and replayed three times, in a
    child process of its own (tools/capture_probe.py, case operator_host_seed)."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "capture_probe.py"), "operator_host_seed"],
                       capture_output=True, text=True, timeout=200)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-2000:]
    assert "exit 0" in p.stdout and "every replay drew new random numbers: True" in p.stdout
