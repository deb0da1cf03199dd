#!/usr/bin/env python3
"""Does node ORDER change what the gathers cost?  (raise the L2 reuse of hub
rows.)  The products-shaped graph is relabelled -- natural R-MAT ids, degree-descending ids (hubs
contiguous at the front), or a random permutation -- and the two gather kernels are timed on each
(library timer); run under tools/profile.sh for FETCH_SIZE / TCC hit counters per labelling:
    python tools/relabel_probe.py natural|degree|random [workload]
The operator's results are the same up to the permutation (rows and columns move together)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("DGNN_TIMING", "1")
import bench  # noqa: E402
from dgnn_amd import _lib  # noqa: E402
from dgnn_amd.native import fused_gatconv as fg  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "natural"
workload = sys.argv[2] if len(sys.argv) > 2 else "products"
dev = torch.device("cuda:0")
n, target, scale = bench.WORKLOADS[workload]
keys = bench.rmat_edges(n, target, scale, dev, 0)
if mode != "natural":
    dst, src = keys // n, keys % n
    if mode == "degree":
        deg = torch.bincount(dst, minlength=n) + torch.bincount(src, minlength=n)
        order = torch.argsort(deg, descending=True, stable=True)          # order[new] = old
    else:
        order = torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))
    new_of_old = torch.empty_like(order)
    new_of_old[order] = torch.arange(n, device=dev)
    keys = torch.sort(new_of_old[dst] * n + new_of_old[src]).values
    del dst, src, order, new_of_old
g = bench.graph_from_keys(keys, n, dev)
del keys
H, F = 4, 64
gen = torch.Generator(device=dev); gen.manual_seed(0)
x = torch.rand((n, H, F), device=dev, generator=gen) - 0.4
ar = torch.rand((n, H), device=dev, generator=gen) - 0.6
ac = torch.rand((n, H), device=dev, generator=gen) - 0.4
go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5
kf = kb = 0.0
reps = int(os.environ.get("DGNN_PROFILE_REPS", "5"))
for i in range(reps + 1):
    out, emax, esum, emask = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0)
    fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum, emask,
                    x, ar, ac, go, out_feat=out)
    torch.cuda.synchronize()
    if i > 0:
        kf += _lib.last_kernel_ms(0)
        kb += _lib.last_kernel_ms(1)
print(json.dumps({"labelling": mode, "workload": workload, "fwd_gather_ms": round(kf / reps, 3),
                  "bwd_gather_ms": round(kb / reps, 3)}), flush=True)
