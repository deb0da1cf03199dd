#!/usr/bin/env python3
"""How many distinct remote rows does each rank of a row-partitioned products-shaped graph need?
(decides all-gather vs halo-only all-to-all in dgnn_amd/dist.py)"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from dgnn_amd.dist import partition_bounds
dev = torch.device("cuda:0")
labels = sys.argv[1] if len(sys.argv) > 1 else "scrambled"      # or "natural"
g = bench.build_graph("products", dev, labels=labels)
n = g["n"]
print("node labels:", labels)
for world in (2, 4, 8):
    b = partition_bounds(g["row_ptr"], world)
    rows = []
    for p in range(world):
        lo, hi = b[p], b[p + 1]
        e0, e1 = int(g["row_ptr"][lo]), int(g["row_ptr"][hi])
        c0, c1 = int(g["col_ptr"][lo]), int(g["col_ptr"][hi])
        uf = torch.unique(g["col_ind"][e0:e1])
        ub = torch.unique(g["row_ind"][c0:c1])
        rf = int(((uf < lo) | (uf >= hi)).sum()); rb = int(((ub < lo) | (ub >= hi)).sum())
        rows.append((hi - lo, e1 - e0, c1 - c0, rf, rb))
    print(f"world={world}  n={n}")
    for p, (nl, ef, eb, rf, rb) in enumerate(rows):
        print(f"  rank {p}: nodes {nl:8d} ({nl/n:5.1%}) fwd_edges {ef:9d} bwd_edges {eb:9d} remote_src_rows {rf:8d} ({rf/(n-nl):5.1%} of remote) remote_dst_rows {rb:8d} ({rb/(n-nl):5.1%})")
