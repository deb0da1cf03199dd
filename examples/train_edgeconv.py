#!/usr/bin/env python3
"""DGCNN classifier on the fused EdgeConv layer with the k-NN graph rebuilt in feature space
before every layer (script/train/train_edgeconv.py: model :12-70, loss :73-79, CSV :288) on
synthetic ModelNet40-shaped point clouds (batch x 2048 points x 3, 40 classes).  The DGL
KNNGraph of the reference is replaced by dgnn_amd.layers.KNNGraph (csrc/knn.hip).

Under ``python -m torch.distributed.run --nproc-per-node N examples/train_edgeconv.py ...`` the
batch is split over N GPUs, whole clouds per rank (dgnn_amd.dist.CloudDataParallel: the k-NN
graph never crosses clouds, so only the weight gradients are all-reduced over RCCL)."""
import argparse
import os

import torch
import torch.nn.functional as F
from torch import nn

from common import accuracy, append_csv, max_memory_mb, time_epochs
from dgnn_amd.layers import EdgeConv, KNNGraph


class Model(nn.Module):
    def __init__(self, k, feature_dims, emb_dims, output_classes, input_dims=3, dropout_prob=0.5):
        super().__init__()
        self.k = k
        self.nng = KNNGraph(k)
        dims = [input_dims] + list(feature_dims)
        self.conv = nn.ModuleList(EdgeConv(dims[i], dims[i + 1]) for i in range(len(feature_dims)))
        self.proj = nn.Linear(sum(feature_dims), emb_dims[0])
        # first embedding takes max- and mean-pooled features side by side
        ins = [emb_dims[0] * 2] + list(emb_dims[1:-1])
        self.embs = nn.ModuleList(nn.Linear(i, o) for i, o in zip(ins, emb_dims[1:]))
        self.bn_embs = nn.ModuleList(nn.BatchNorm1d(o) for o in emb_dims[1:])
        self.dropouts = nn.ModuleList(nn.Dropout(dropout_prob) for _ in emb_dims[1:])
        self.proj_output = nn.Linear(emb_dims[-1], output_classes)

    def forward(self, x):
        batch, points, _ = x.shape
        h, hs = x, []
        for conv in self.conv:
            src_ind = self.nng(h)                                   # graph in the current feature space
            h = conv(self.k, src_ind, h.reshape(batch * points, -1))
            h = F.leaky_relu(h, 0.2).view(batch, points, -1)
            hs.append(h)
        h = self.proj(torch.cat(hs, 2))
        h = torch.cat([h.max(1).values, h.mean(1)], 1)
        for emb, bn, drop in zip(self.embs, self.bn_embs, self.dropouts):
            h = drop(F.leaky_relu(bn(emb(h)), 0.2))
        return self.proj_output(h)


def compute_loss(logits, y, eps=0.2):
    """Label-smoothed cross entropy (train_edgeconv.py:73-79)."""
    n = logits.shape[1]
    one_hot = torch.zeros_like(logits).scatter_(1, y.view(-1, 1), 1)
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / (n - 1)
    return -(one_hot * F.log_softmax(logits, 1)).sum(1).mean()


def main():
    ap = argparse.ArgumentParser(description="DGCNN on dgnn_amd")
    ap.add_argument("--gpu", type=int, default=0)
    ap.add_argument("--batch-size", type=int, default=32)
    ap.add_argument("--num-points", type=int, default=2048)
    ap.add_argument("--k", type=int, default=20)
    ap.add_argument("--emb-dims", default="512,512,256",
                    help="widths of the classifier head; default = the reference's Model(..., [512, 512, 256], 40), "
                         "script/train/train_edgeconv.py:244")
    ap.add_argument("--n-epochs", type=int, default=20, help="timed iterations over one synthetic batch")
    ap.add_argument("--lr", type=float, default=0.1)
    ap.add_argument("--dropout", type=float, default=0.5)
    ap.add_argument("--output", default=None)
    args = ap.parse_args()
    print(args)

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        args.gpu = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(args.gpu)
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{args.gpu}"))
    device = torch.device(f"cuda:{args.gpu}")
    torch.cuda.set_device(device)
    torch.manual_seed(0)
    x = torch.randn((args.batch_size, args.num_points, 3), device=device)
    y = torch.randint(0, 40, (args.batch_size,), device=device)
    emb_dims = [int(v) for v in args.emb_dims.split(",")]
    model = Model(args.k, [64, 64, 128, 256], emb_dims, 40, dropout_prob=args.dropout).to(device)   # :244
    from dgnn_amd.dist import CloudDataParallel
    dp = CloudDataParallel(model, args.batch_size, world, rank)
    x, y = dp.local(x).contiguous(), dp.local(y).contiguous()      # this rank's clouds
    opt = torch.optim.SGD(model.parameters(), lr=args.lr, momentum=0.9, weight_decay=1e-4, fused=True)
    losses = []

    def train_iter():
        loss = compute_loss(model(x), y)
        opt.zero_grad()
        loss.backward()
        dp.reduce_gradients()
        opt.step()
        losses.append(loss.detach())

    def infer_iter():
        with torch.no_grad():
            return model(x)

    model.train()
    train_time = time_epochs(train_iter, args.n_epochs)
    model.eval()
    inference_time = time_epochs(infer_iter, args.n_epochs, warmup=0)
    acc = accuracy(infer_iter(), y)
    if world > 1:
        t = torch.tensor([train_time, inference_time], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)               # the step ends when the slowest rank ends
        train_time, inference_time = float(t[0]), float(t[1])
    if rank == 0:
        print(f"loss {losses[0].item():.4f} -> {losses[-1].item():.4f}; accuracy on the batch {acc:.4f}")
        tag = f"{args.batch_size} {args.k}" + (f" dp{world}" if world > 1 else "")
        append_csv(args.output, f"train_edgeconv_dgnn_amd,{tag},{train_time}s,"
                                f"{inference_time}s,{max_memory_mb(device)}MB,{acc}")
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
