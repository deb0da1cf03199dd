#!/usr/bin/env python3
"""Host-side cost per call on a Cora-sized graph (launch-bound regime): wall time per call of
the native wrappers and of a whole layer, measured over many back-to-back calls."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from common import synthetic_citation  # noqa: E402
from dgnn_amd.layers import GATConv  # noqa: E402
from dgnn_amd.native import fused_gatconv as fg  # noqa: E402

dev = torch.device("cuda:0")
d = synthetic_citation("cora", dev)
rp, ci, cp, ri, pm = d["graph"]
n, H, F = d["n"], 4, 64
x = torch.randn((n, H, F), device=dev)
ar, ac = torch.randn((n, H), device=dev), torch.randn((n, H), device=dev)
go = torch.randn((n, H, F), device=dev)


def wall(fn, iters=300):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e6


out, emax, esum, emask = fg.gat_forward(ar, ac, rp, ci, 0.2, x, 0.0)
print("gat_inference  us/call", round(wall(lambda: fg.gat_inference(ar, ac, rp, ci, 0.2, x)), 1))
print("gat_forward    us/call", round(wall(lambda: fg.gat_forward(ar, ac, rp, ci, 0.2, x, 0.0)), 1))
print("gat_backward   us/call", round(wall(lambda: fg.gat_backward(0.2, 0.0, rp, ci, cp, ri, pm, emax, esum, emask, x, ar, ac, go, out_feat=out)), 1))
print("torch.empty    us/call", round(wall(lambda: torch.empty((n, H, F), device=dev)), 1))
print("matmul 1433->256 us/call", round(wall(lambda: d["features"] @ torch.empty((1433, 256), device=dev)), 1))
layer = GATConv(1433, 64, 4).to(dev).eval()
with torch.no_grad():
    print("GATConv layer eval us/call", round(wall(lambda: layer(rp, ci, cp, ri, pm, d["features"])), 1))
