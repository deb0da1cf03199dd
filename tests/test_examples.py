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
