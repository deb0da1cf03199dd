#!/usr/bin/env python3
"""The exchange-plan model of dgnn_amd.dist over chunk shapes / counts / assumed link rates on the
products-shaped graph (rank 0's share; GPU box, no peers needed):
    python tools/plan_model.py > gpurun_out/plan_model.jsonl
One line per (world, shape, chunks): PartitionedGAT.expected_ms at 64 and 100 GB/s per link."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dgnn_amd.dist as D  # noqa: E402

dev = torch.device("cuda:0")
g = bench.build_graph("products", dev, seed=0, labels="scrambled")
G = {k: g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind")}
for world in (8, 4, 2):
    for shape, chunks in (("geometric", 4), ("geometric", 6), ("equal", 4), ("equal", 8), ("geometric", 1)):
        os.environ["DGNN_DIST_CHUNK_SHAPE"] = shape
        r = D.PartitionedGAT(G, world, 0, dev, fake_peers=True, mode="halo", chunks=chunks)
        res = {}
        for link in (64.0, 100.0):
            D.LINK_GBS = link
            e = r.expected_ms(4, 64)
            res[f"{link:g} GB/s"] = {"ms_per_step": e["ms_per_step"], "fwd_wire_ms": e["forward"]["wire_ms"],
                                     "fwd_compute_ms": e["forward"]["compute_ms"]}
        print(json.dumps({"world": world, "shape": shape, "chunks": chunks, "expected": res,
                          "halo_fraction": r.plan["halo_fraction_min_over_ranks"],
                          "first_chunk_share_of_halo": round(r.f_side.halos[0].n_needed / max(1, sum(h.n_needed for h in r.f_side.halos)), 3)}),
              flush=True)
        del r
        torch.cuda.empty_cache()
