#!/usr/bin/env python3
"""The program behind the per-operator rocprofv3 passes (tools/profile.sh <tag> <name>
tools/profile_ops.py <name>): a few launches of ONE operator family at its BASELINE.json shape.
  gmm_pubmed / gmm_reddit   fused GMMConv fwd + one-pass bwd (K = 3, D = 2, F = 64)
  edgeconv                  fused EdgeConv fwd / bwd (atomic + deterministic) + k-NN, 32 x 2048, k = 20,
                            the four DGCNN widths
  sddmm_spmm                mhsddmm + spmm_csr on the Reddit shape, 4 x 64
  gat_1x64_products / gat_8x8_products / gat_1x64_reddit   narrow fused GAT fwd + bwd
  gatneg_<HxF>_<workload>                                  the same on the operator's negative-score path
  layer_products            whole GATConv layer (in = 100, 4 x 64, ELU, bias) fwd + bwd and an eval forward:
                            what the score / epilogue kernels around the operator cost"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402

dev = torch.device("cuda:0")
REPS = int(os.environ.get("DGNN_PROFILE_REPS", "3"))


def gmm(workload, summed=False):
    """summed: the layer's form, out.sum(1): two-step (operator, torch sum; DGNN_GMM_TWO_STEP=1) or folded."""
    from dgnn_amd.operators import GmmConvFuse, GmmConvFuseSum
    if workload == "pubmed":
        from util import random_graph
        g = random_graph(19717, 88648, 0)
        g = {k: (torch.from_numpy(v).to(dev) if hasattr(v, "shape") else v) for k, v in g.items()}
    else:
        g = bench.build_graph(os.environ.get("DGNN_PROFILE_GMM_GRAPH", "reddit"), dev)    # or "products": short columns
    K, D, F = 3, 2, int(os.environ.get("DGNN_PROFILE_GMM_F", "64"))     # 16: the reference script's default width
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    x = torch.randn((g["n"], K, F), device=dev, generator=gen).requires_grad_()
    ps = torch.rand((g["nnz"], D), device=dev, generator=gen).requires_grad_()
    mu = (0.1 * torch.randn((K, D), device=dev, generator=gen)).requires_grad_()
    sg = torch.ones((K, D), device=dev).requires_grad_()
    go = torch.randn((g["n"], K, F), device=dev, generator=gen)
    gs = torch.randn((g["n"], F), device=dev, generator=gen)
    G5 = [g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for i in range(REPS + 1):
        if i == 1:
            ev[0].record()
        x.grad = ps.grad = mu.grad = sg.grad = None
        if not summed:
            GmmConvFuse(*G5, x, ps, mu, sg).backward(go)
        elif os.environ.get("DGNN_GMM_TWO_STEP") == "1":
            GmmConvFuse(*G5, x, ps, mu, sg).sum(1).backward(gs)
        else:
            GmmConvFuseSum(*G5, x, ps, mu, sg).backward(gs)
    ev[1].record()
    torch.cuda.synchronize()
    print(f"gmm {workload} summed={summed} two_step={os.environ.get('DGNN_GMM_TWO_STEP') == '1'}: "
          f"{ev[0].elapsed_time(ev[1]) / REPS:.3f} ms per fwd+bwd", flush=True)


def edgeconv():
    from dgnn_amd.native import fused_edgeconv as fe
    from dgnn_amd.native.knn import knn_graph
    batch, pts, k = 32, 2048, 20
    m = batch * pts
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    for dim, f in ((3, 64), (64, 64), (64, 128), (128, 256)):
        x = torch.randn((batch, pts, dim), device=dev, generator=gen)
        hs = torch.randn((m, f), device=dev, generator=gen)
        hd = torch.randn((m, f), device=dev, generator=gen)
        go = torch.randn((m, f), device=dev, generator=gen)
        for _ in range(REPS):
            src = knn_graph(x, k)
            out, idx_f, idx_i = fe.edgeconv_forward(k, src, hs, hd, return_int_idx=True)
            fe.edgeconv_backward(go, idx_i)
            fe.edgeconv_backward(go, idx_i, src_ind=src)


def sddmm_spmm():
    from dgnn_amd.native import mhsddmm, spmm
    g = bench.build_graph("reddit", dev)
    H, F = 4, 64
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    x = torch.rand((g["n"], H, F), device=dev, generator=gen) - 0.4
    go = torch.rand((g["n"], H, F), device=dev, generator=gen) - 0.5
    val = torch.rand((g["nnz"], H), device=dev, generator=gen)
    for _ in range(REPS):
        spmm.spmm_csr(g["row_ptr"], g["col_ind"], val, x)
        mhsddmm.mhsddmm(g["row_ptr"], g["col_ind"], go, x)


def gat(H, F, workload, neg=False):
    """neg: the operator's path (forward writes the negative-score pair, the backward reads it)."""
    from dgnn_amd.native import fused_gatconv as fg
    g = bench.build_graph(workload, dev)
    n = g["n"]
    x = torch.rand((n, H, F), device=dev) - 0.4
    ar, ac = torch.rand((n, H), device=dev) - 0.6, torch.rand((n, H), device=dev) - 0.4
    go = torch.rand((n, H, F), device=dev) - 0.5
    for _ in range(REPS):
        on = an = None
        if neg:
            out, emax, esum, emask, on, an = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=True)
        else:
            out, emax, esum, emask = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0)
        fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum,
                        emask, x, ar, ac, go, out_feat=out, out_neg=on, a_neg=an)


def layer(workload):
    import torch.nn.functional as F
    from dgnn_amd.layers import GATConv
    g = bench.build_graph(workload, dev)
    graph = [g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    torch.manual_seed(0)
    lay = GATConv(100, 64, 4, activation=F.elu).to(dev)
    feat = torch.randn((g["n"], 100), device=dev, requires_grad=True)
    go = torch.randn((g["n"], 4, 64), device=dev)
    for _ in range(REPS):
        lay.zero_grad(); feat.grad = None
        lay.train()
        lay(*graph, feat).backward(go)
        lay.eval()
        with torch.no_grad():
            lay(*graph, feat)


if __name__ == "__main__":
    w = sys.argv[1]
    if w.startswith("gmmsum_"):
        gmm(w[7:], summed=True)
    elif w.startswith("gmm_"):
        gmm(w[4:])
    elif w == "edgeconv":
        edgeconv()
    elif w == "sddmm_spmm":
        sddmm_spmm()
    elif w.startswith("layer_"):
        layer(w[6:])
    elif w.startswith("gat_") or w.startswith("gatneg_"):
        kind, hf, wl = w.split("_")
        h, f = hf.split("x")
        gat(int(h), int(f), wl, neg=kind == "gatneg")
    else:
        raise SystemExit(f"unknown operator set {w}")
    torch.cuda.synchronize()
