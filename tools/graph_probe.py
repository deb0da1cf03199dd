#!/usr/bin/env python3
"""Probe: one GAT inference epoch and one training epoch of examples/train_gatconv.py on a
Cora-sized graph, replayed from a HIP graph (torch.cuda.CUDAGraph) against eager launches."""
import os
import sys
import time

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "examples"))
from common import synthetic_citation  # noqa: E402
from train_gatconv import Net  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cora"
what = sys.argv[2] if len(sys.argv) > 2 else "infer"
dev = torch.device("cuda:0")
torch.manual_seed(0)
data = synthetic_citation(name, dev)
model = Net(data["graph"], 1, data["n_feats"], 64, data["n_classes"], [4, 1], feat_drop=0.0).to(dev)
feats, labels = data["features"], data["labels"]
train_idx = data["train_mask"].nonzero().squeeze(1)
train_labels = labels[train_idx]
loss_fcn = nn.CrossEntropyLoss()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=5e-4, capturable=True)


def wall(fn, iters=300):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / iters * 1e3


if what == "infer":
    model.eval()

    def step():
        with torch.no_grad():
            return model(feats)
else:
    model.train()

    def step():
        loss = loss_fcn(model(feats)[train_idx], train_labels)
        opt.zero_grad(set_to_none=False)
        loss.backward()
        opt.step()
        return loss

print(what, name, "eager ms", round(wall(step), 4), flush=True)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print("capturing", flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
torch.cuda.synchronize()
print("captured", flush=True)
ref = step().detach().clone() if what == "infer" else None
g.replay()
torch.cuda.synchronize()
if ref is not None:
    print("replay equals eager:", bool(torch.equal(out, ref)))
print(what, name, "graph replay ms", round(wall(g.replay), 4), flush=True)
