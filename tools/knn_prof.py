"""Profile program: filtered k-NN at 32 x 2048 points, k = 20, dims 64 and 128 (tools/profile.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgnn_amd  # noqa: E402,F401
from dgnn_amd.native import knn  # noqa: E402

dev = torch.device("cuda:0")
args = sys.argv[1:] or ["32", "2048", "20", "64"]
batch, points, k, dim = (int(v) for v in args)
x = torch.randn((batch, points, dim), device=dev)
for _ in range(5):
    knn.knn_graph(x, k)
torch.cuda.synchronize()
