#!/usr/bin/env python3
"""SpMM (GE-SpMM entry point) and SDDMM on narrow head x feature shapes, products-shaped graph, scrambled labels:
   DGNN_FWD_L=... python tools/spmm_narrow.py 1x64 4x16 ...   (group-size A/B for the MODE_VAL forward)"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bench_ops import timeit  # noqa: E402
from dgnn_amd.native import mhsddmm, spmm  # noqa: E402

dev = torch.device("cuda:0")
g = bench.build_graph("products", dev, labels="scrambled")
n, nnz = g["n"], g["nnz"]
gen = torch.Generator(device=dev); gen.manual_seed(0)
for shape in sys.argv[1:] or ["1x64", "4x16", "2x32", "1x16", "2x64", "4x64"]:
    H, F = (int(v) for v in shape.split("x"))
    x = torch.rand((n, H, F), device=dev, generator=gen) - 0.4
    go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5
    val = torch.rand((nnz, H), device=dev, generator=gen)
    ts = timeit(lambda: spmm.spmm_csr(g["row_ptr"], g["col_ind"], val, x), iters=5, warm=2)
    td = timeit(lambda: mhsddmm.mhsddmm(g["row_ptr"], g["col_ind"], go, x), iters=5, warm=2)
    print(json.dumps({"shape": shape, "spmm_ms": round(ts, 3), "sddmm_ms": round(td, 3)}), flush=True)
    del x, go, val
