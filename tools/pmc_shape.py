#!/usr/bin/env python3
"""One forward + backward of the fused GAT on the products-shaped graph at a given (H, F):
the program rocprofv3 --pmc passes are run on (tools/pmc_shape.sh)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dgnn_amd.native import fused_gatconv as fg  # noqa: E402

H, F = int(sys.argv[1]), int(sys.argv[2])
dev = torch.device("cuda:0")
g = bench.build_graph("products", dev)
n = g["n"]
x = torch.rand((n, H, F), device=dev) - 0.4
ar, ac = torch.rand((n, H), device=dev) - 0.6, torch.rand((n, H), device=dev) - 0.4
go = torch.rand((n, H, F), device=dev) - 0.5
neg = os.environ.get("DGNN_BENCH_NEG", "1") == "1"       # the operator's path: negative-score pair
for _ in range(2):
    out, emax, esum, emask, on, an = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=True)
    if not neg:
        on = an = None
    fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum, emask,
                    x, ar, ac, go, out_feat=out, out_neg=on, a_neg=an)
torch.cuda.synchronize()
