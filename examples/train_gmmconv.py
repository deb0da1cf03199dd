#!/usr/bin/env python3
"""MoNet on the fused GMMConv layer (script/train/train_gmmconv.py: model :11-59, pseudo
coordinates 1/sqrt(in-degree) of both end points :120-122, CSV :190) on a synthetic graph."""
import argparse

import torch
from torch import nn

from common import (accuracy, append_csv, capture_step, cross_entropy, max_memory_mb, synthetic_citation,
                    time_epochs)
from dgnn_amd.layers import GMMConv


class PseudoLinear(nn.Linear):
    """nn.Linear(2, dim) applied to one row per EDGE (train_gmmconv.py:38): same parameters and
    state_dict keys, but evaluated as a broadcast multiply-add -- as a GEMM with K = 2 it is a
    114 M x 2 x 2 problem that the BLAS library runs at a fraction of a percent of its peak
    (131 ms per epoch on the Reddit shape, half of the whole epoch)."""

    def forward(self, x):
        if self.in_features > 4:
            return super().forward(x)
        out = x[:, 0:1] * self.weight[:, 0]
        for i in range(1, self.in_features):
            out = out + x[:, i:i + 1] * self.weight[:, i]
        return out + self.bias if self.bias is not None else out


class MoNet(nn.Module):
    def __init__(self, graph, in_feats, n_hidden, out_feats, n_layers, dim, n_kernels, dropout):
        super().__init__()
        self.graph = graph
        dims = [in_feats] + [n_hidden] * n_layers + [out_feats]
        self.layers = nn.ModuleList(GMMConv(dims[i], dims[i + 1], dim, n_kernels)
                                    for i in range(n_layers + 1))
        self.pseudo_proj = nn.ModuleList(nn.Sequential(PseudoLinear(2, dim), nn.Tanh())
                                         for _ in range(n_layers + 1))
        self.dropout = nn.Dropout(dropout)

    def forward(self, h, pseudo):
        for i, (layer, proj) in enumerate(zip(self.layers, self.pseudo_proj)):
            if i:
                h = self.dropout(h)
            h = layer(*self.graph, h, proj(pseudo))
        return h


def main():
    ap = argparse.ArgumentParser(description="MoNet on dgnn_amd")
    ap.add_argument("--dataset", default="pubmed", choices=["cora", "citeseer", "pubmed", "reddit"])
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--lr", type=float, default=1e-2)
    ap.add_argument("--weight-decay", type=float, default=5e-5)
    ap.add_argument("--n-epochs", type=int, default=200)
    ap.add_argument("--n-hidden", type=int, default=16)
    ap.add_argument("--n-layers", type=int, default=1)
    ap.add_argument("--pseudo-dim", type=int, default=2)
    ap.add_argument("--n-kernels", type=int, default=3)
    ap.add_argument("--output", default=None)
    ap.add_argument("--hip-graph", action="store_true",
                    help="capture the training epoch into a HIP graph and replay it (launch-bound small graphs)")
    args = ap.parse_args()
    if args.hip_graph and (args.dropout != 0):
        # checked before anything touches the GPU: capturing the dropout generators aborted the
        # process and faulted the card on this ROCm build
        ap.error("--hip-graph needs --dropout 0: random-number ops are not captured")
    print(args)

    device = torch.device(f"cuda:{args.gpu}")
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    data = synthetic_citation(args.dataset, device)
    row_ptr, col_ind = data["graph"][0].long(), data["graph"][1].long()
    # pseudo coordinates per edge, in CSR nnz order: 1/sqrt(in-degree) of source and
    # destination (train_gmmconv.py:120-122)
    deg = (row_ptr[1:] - row_ptr[:-1]).float()
    dst = torch.repeat_interleave(torch.arange(data["n"], device=device), row_ptr[1:] - row_ptr[:-1])
    pseudo = torch.stack([deg[col_ind].rsqrt(), deg[dst].rsqrt()], 1).contiguous()

    model = MoNet(data["graph"], data["n_feats"], args.n_hidden, data["n_classes"], args.n_layers,
                  args.pseudo_dim, args.n_kernels, args.dropout).to(device)
    loss_fcn = cross_entropy
    optimizer = torch.optim.Adam(model.parameters(), lr=args.lr, weight_decay=args.weight_decay,
                                 capturable=args.hip_graph, fused=True)   # one launch instead of a foreach chain
    feats, labels = data["features"], data["labels"]
    train_idx = data["train_mask"].nonzero().squeeze(1)      # index list: no host sync inside the epoch
    train_labels = labels[train_idx]
    losses = []

    def train_step():
        loss = loss_fcn(model(feats, pseudo)[train_idx], train_labels)
        optimizer.zero_grad(set_to_none=not args.hip_graph)
        loss.backward()
        optimizer.step()
        return loss.detach()

    def train_epoch():
        losses.append(train_step())

    def infer_epoch():
        with torch.no_grad():
            return model(feats, pseudo)

    model.train()
    if args.hip_graph:
        losses.append(train_step().clone())
        replay, static_loss = capture_step(train_step)
        train_time = time_epochs(replay, args.n_epochs)
        losses.append(static_loss.clone())
    else:
        train_time = time_epochs(train_epoch, args.n_epochs)
    model.eval()
    inference_time = time_epochs(infer_epoch, args.n_epochs, warmup=0)
    acc = accuracy(infer_epoch()[data["test_mask"]], labels[data["test_mask"]])
    print(f"loss {losses[0].item():.4f} -> {losses[-1].item():.4f}; test accuracy {acc:.4f} (random labels)")
    append_csv(args.output, "train_GMM_dgnn_amd,{} pseudo_dim={} n_kernels={} hidden_dim={}{},{:f}s,{:f}s,{}MB".format(
        args.dataset, args.pseudo_dim, args.n_kernels, args.n_hidden, " hip-graph" if args.hip_graph else "",
        train_time, inference_time,
        max_memory_mb(device)))


if __name__ == "__main__":
    main()
