"""k-NN graph timing, exhaustive kernel against the filtered path (DGNN_KNN_FILTER=0/1), and the share
of query blocks the filter handed back to the exhaustive kernel.  python tools/knn_bench.py"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import dgnn_amd  # noqa: E402,F401
from dgnn_amd.native import knn  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


for batch, points, k in ((32, 2048, 20), (64, 1024, 40)):
    for dim in (3, 32, 64, 128):
        x = torch.randn((batch, points, dim), device=dev)
        if os.environ.get("DGNN_KNN_BENCH_DATA") == "features":     # what an activation leaves: positive, far from 0
            x = torch.nn.functional.leaky_relu(x + 1.0, 0.2) + 0.5
        res = {}
        for flt in ("0", "1"):
            os.environ["DGNN_KNN_FILTER"] = flt
            res[flt] = knn.knn_graph(x, k)
            res["ms" + flt] = timeit(lambda: knn.knn_graph(x, k))
        same = bool(torch.equal(res["0"], res["1"]))
        print(json.dumps({"batch": batch, "points": points, "k": k, "dim": dim, "exhaustive_ms": round(res["ms0"], 3),
                          "filtered_ms": round(res["ms1"], 3), "same_ids": same}), flush=True)
