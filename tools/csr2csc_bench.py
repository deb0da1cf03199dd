"""Device CSR -> CSC (+ permute) timing at the bench shapes.  python tools/csr2csc_bench.py [products|reddit]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from dgnn_amd.native import format_conversion  # noqa: E402

dev = torch.device("cuda:0")
for w in (sys.argv[1:] or ["products", "reddit"]):
    g = bench.build_graph(w, dev)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2):
        out = format_conversion.csr2csc_int(g["row_ptr"], g["col_ind"])
    torch.cuda.synchronize()
    a.record()
    for _ in range(5):
        out = format_conversion.csr2csc_int(g["row_ptr"], g["col_ind"])
    b.record()
    torch.cuda.synchronize()
    same = all(bool(torch.equal(x, y)) for x, y in zip(out, (g["col_ptr"], g["row_ind"], g["permute"])))
    print(json.dumps({"op": "csr2csc_int", "workload": w, "nnz": g["nnz"], "ms": round(a.elapsed_time(b) / 5, 3),
                      "same_as_graph": same}), flush=True)
    del g, out
