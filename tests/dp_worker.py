"""Worker for the multi-process tests of dgnn_amd.dist.CloudDataParallel (spawned by
test_dist_edgeconv.py): data parallel over point clouds for the EdgeConv / DGCNN path.

mode "cpu": gloo, CPU tensors, float64; the model is a pure-torch statement of the DGCNN layer
            stack (k-NN graph, EdgeConv as gather + max, pooling, linear head): the test is
            about the split, the parameter broadcast and the weighted gradient all-reduce.
mode "gpu": gloo, every rank on cuda:0, dgnn_amd.layers.EdgeConv / KNNGraph (the HIP kernels).
mode "nccl": the real thing where the box has a GPU per rank: rank r on cuda:r, RCCL collectives.
Rank r seeds its parameters with 100 + r, so the broadcast from rank 0 is really needed.
Each rank writes its reduced gradients and its parameters to <out_dir>/rank<r>.npz.
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (ROOT, HERE):
    if p not in sys.path:
        sys.path.insert(0, p)


GPU = "cuda:0"           # the device of the GPU modes ("nccl": cuda:<rank>, set in run())


def knn_torch(x, k):
    """[B,P,D] -> src_ind [B*P*k] (global point ids, self included, ascending distance)."""
    b, p, _ = x.shape
    d = torch.cdist(x, x)
    idx = d.topk(k, dim=2, largest=False).indices                  # [B,P,k]
    return (idx + (torch.arange(b, device=x.device) * p).view(b, 1, 1)).reshape(-1)


class TorchEdgeConv(nn.Module):
    """edgeconv_layer.py:34-41 with the fused operator written as gather + max."""

    def __init__(self, i, o):
        super().__init__()
        self.theta, self.phi = nn.Linear(i, o, bias=False), nn.Linear(i, o, bias=False)

    def forward(self, k, src_ind, feat):
        h_src = self.theta(feat)
        h_dst = self.phi(feat) - h_src
        return (h_dst.unsqueeze(1) + h_src[src_ind.long()].view(feat.size(0), k, -1)).max(1).values


class SmallDGCNN(nn.Module):
    """Two EdgeConv layers with the graph rebuilt in feature space, max+mean pooling, a linear
    head; no batch statistics, so data parallel == full batch exactly (up to summation order)."""

    def __init__(self, k, conv_cls, knn):
        super().__init__()
        self.k, self.knn = k, knn
        self.conv = nn.ModuleList([conv_cls(3, 16), conv_cls(16, 24)])
        self.head = nn.Linear(2 * 40, 5)
        self.unused = nn.Parameter(torch.zeros(3))          # never touched by forward

    def forward(self, x):
        b, p, _ = x.shape
        h, hs = x, []
        for conv in self.conv:
            src = self.knn(h)
            h = torch.tanh(conv(self.k, src, h.reshape(b * p, -1))).view(b, p, -1)
            hs.append(h)
        h = torch.cat(hs, 2)
        return self.head(torch.cat([h.max(1).values, h.mean(1)], 1))


def build(mode, k, seed):
    torch.manual_seed(seed)
    if mode == "cpu":
        return SmallDGCNN(k, TorchEdgeConv, lambda h: knn_torch(h, k)).double()
    from dgnn_amd.layers import EdgeConv, KNNGraph
    return SmallDGCNN(k, EdgeConv, KNNGraph(k)).to(GPU)


def data(mode, batch, points):
    g = torch.Generator().manual_seed(7)
    x = torch.randn((batch, points, 3), generator=g, dtype=torch.float64)
    y = torch.randint(0, 5, (batch,), generator=g)
    if mode == "cpu":
        return x, y
    return x.float().to(GPU), y.to(GPU)


def grads_of(model):
    return {n: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().cpu().double().numpy()
            for n, p in model.named_parameters()}


def run(rank, world, mode, port, out_dir, batch, points, k):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    from datetime import timedelta
    if mode == "nccl":
        global GPU
        GPU = f"cuda:{rank}"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(GPU),
                                timeout=timedelta(seconds=120))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world, timeout=timedelta(seconds=120))
    from dgnn_amd.dist import CloudDataParallel
    model = build(mode, k, seed=100 + rank)
    dp = CloudDataParallel(model, batch, world, rank)
    x, y = data(mode, batch, points)
    loss = nn.functional.cross_entropy(dp(dp.local(x)), dp.local(y))      # mean over the local clouds
    loss.backward()
    dp.reduce_gradients()
    out = {"g:" + n: v for n, v in grads_of(model).items()}
    out.update({"p:" + n: p.detach().cpu().double().numpy() for n, p in model.named_parameters()})
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), lo=dp.lo, hi=dp.hi, **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    a = sys.argv
    run(int(a[1]), int(a[2]), a[3], int(a[4]), a[5], int(a[6]), int(a[7]), int(a[8]))
