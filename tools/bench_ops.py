#!/usr/bin/env python3
"""Secondary operator benchmarks on one GPU (BASELINE.json configs 2-4): fused EdgeConv on a
ModelNet40-shaped batch, fused GMMConv on a PubMed-shaped graph, SpMM / SDDMM / GAT on a
Reddit-shaped R-MAT.  Prints one JSON line per operator with algorithmic GB/s."""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

dev = torch.device("cuda:0")


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(iters):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) / iters


def edgeconv(batch=32, pts=2048, k=20):
    from dgnn_amd.native import fused_edgeconv as fe
    m = batch * pts
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    # k random neighbours inside each cloud (the k-NN graph itself is DGL's job in the reference)
    base = (torch.arange(m, device=dev) // pts * pts)[:, None]
    src = (base + torch.randint(0, pts, (m, k), device=dev, generator=gen)).to(torch.int32).reshape(-1).contiguous()
    for f in (64, 64, 128, 256):
        hs = torch.randn((m, f), device=dev, generator=gen)
        hd = torch.randn((m, f), device=dev, generator=gen)
        go = torch.randn((m, f), device=dev, generator=gen)
        out, idx_f, idx_i = fe.edgeconv_forward(k, src, hs, hd, return_int_idx=True)
        tf = timeit(lambda: fe.edgeconv_forward(k, src, hs, hd, return_int_idx=True))
        tb = timeit(lambda: fe.edgeconv_backward(go, idx_i))
        td = timeit(lambda: fe.edgeconv_backward(go, idx_i, src))      # deterministic form (the autograd operator's)
        bf = m * (4 * k + 4 * k * f + 4 * f + 12 * f)
        bb = m * (4 * f + 4 * f + 8 * f + 4 * f)
        print(json.dumps({"op": "edgeconv", "m": m, "k": k, "f": f, "fwd_ms": round(tf, 4), "bwd_ms": round(tb, 4),
                          "bwd_det_ms": round(td, 4), "fwd_GBps": round(bf / tf / 1e6, 1), "bwd_GBps": round(bb / tb / 1e6, 1)}), flush=True)


def knn(batch=32, pts=2048, k=20):
    """k-NN graph of a ModelNet40-shaped batch in the feature spaces DGCNN builds it in
    (script/train/train_edgeconv.py:49: input dims 3, 64, 64, 128)."""
    from dgnn_amd.native.knn import knn_graph
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    for f in (3, 64, 128):
        x = torch.randn((batch, pts, f), device=dev, generator=gen)
        t = timeit(lambda: knn_graph(x, k), iters=5, warm=2)
        pairs = batch * pts * pts
        print(json.dumps({"op": "knn_graph", "batch": batch, "points": pts, "dim": f, "k": k, "ms": round(t, 4),
                          "Gpairs_per_s": round(pairs / t / 1e6, 2),
                          "fp32_TFLOPs": round(pairs * 3 * f / t / 1e9, 2)}), flush=True)


def layer(workload="products"):
    """Whole GATConv layer (projection GEMM + score terms + fused operator + bias/ELU tail),
    forward + backward: the layer as shipped against the same layer written with the
    reference's chain of torch ops around the fused operator (gatconv_layer.py:68-92)."""
    import torch.nn.functional as F
    from dgnn_amd.layers import GATConv
    from dgnn_amd.operators import GATConvFuse
    g = bench.build_graph(workload, dev)
    graph = [g[k] for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    n, in_feats, H, Fo = g["n"], 100, 4, 64
    torch.manual_seed(0)
    lay = GATConv(in_feats, Fo, H, activation=F.elu).to(dev)
    feat = torch.randn((n, in_feats), device=dev, requires_grad=True)
    go = torch.randn((n, H, Fo), device=dev)

    def plain():
        h = torch.matmul(feat, lay.W).view(-1, H, Fo)
        rst = GATConvFuse((lay.attn_l * h).sum(-1), (lay.attn_r * h).sum(-1), *graph, lay.negative_slope, h, 0.0)
        return F.elu(rst + lay.bias.view(-1, H, Fo))

    def step(fn):
        lay.zero_grad(); feat.grad = None
        fn().backward(go)
    t_fused = timeit(lambda: step(lambda: lay(*graph, feat)), iters=5, warm=2)
    t_plain = timeit(lambda: step(plain), iters=5, warm=2)
    print(json.dumps({"op": f"GATConv layer fwd+bwd ({workload}-shaped, in=100, 4x64, ELU)", "fused_ms": round(t_fused, 3),
                      "torch_chain_ms": round(t_plain, 3)}), flush=True)


SHAPES = ((1, 47), (1, 48), (1, 64), (1, 16), (8, 8), (4, 16), (2, 32), (1, 7), (8, 32), (2, 64), (4, 41), (4, 64), (8, 64), (4, 128), (4, 256))


def shapes(workload="products"):
    """Fused GAT forward / backward over head x feature shapes other than the headline 4 x 64:
    output layers (1 head x n_classes) and odd widths."""
    from dgnn_amd.native import fused_gatconv as fg
    g = bench.build_graph(workload, dev, labels=os.environ.get("DGNN_BENCH_LABELS", "natural"))
    n, nnz = g["n"], g["nnz"]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    todo = SHAPES           # DGNN_BENCH_SHAPES="1x44,1x48": another list
    if os.environ.get("DGNN_BENCH_SHAPES"):
        todo = tuple(tuple(int(v) for v in it.split("x")) for it in os.environ["DGNN_BENCH_SHAPES"].split(","))
    for H, F in todo:
        x = torch.rand((n, H, F), device=dev, generator=gen) - 0.4
        ar = torch.rand((n, H), device=dev, generator=gen) - 0.6
        ac = torch.rand((n, H), device=dev, generator=gen) - 0.4
        go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5
        neg = os.environ.get("DGNN_BENCH_NEG", "1") == "1"      # the operator's path: negative-score pair
        out, emax, esum, emask, on, an = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=True)
        if not neg:
            on = an = None
        tf = timeit(lambda: fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0, neg=neg), iters=int(os.environ.get("DGNN_BENCH_ITERS", "3")), warm=1)
        tb = timeit(lambda: fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"],
                                            emax, esum, emask, x, ar, ac, go, out_feat=out, out_neg=on, a_neg=an),
                    iters=int(os.environ.get("DGNN_BENCH_ITERS", "3")), warm=1)
        ab = bench.algorithmic_bytes(n, nnz, H, F, neg=neg and on is not None)
        print(json.dumps({"op": f"gat {workload}", "H": H, "F": F, "fwd_ms": round(tf, 3), "bwd_ms": round(tb, 3),
                          "fwd_GBps": round(ab["fwd"] / tf / 1e6, 1), "bwd_GBps": round(ab["bwd_csc"] / tb / 1e6, 1)}), flush=True)
        del x, go, out


def gmm_pubmed():
    from dgnn_amd.operators import GmmConvFuse
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import random_graph
    g = random_graph(19717, 88648, 0)
    G = {k2: torch.from_numpy(g[k2]).to(dev) for k2 in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")}
    K, D, F = 3, 2, 64
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    x = torch.randn((g["n"], K, F), device=dev, generator=gen).requires_grad_()
    ps = torch.rand((g["nnz"], D), device=dev, generator=gen).requires_grad_()
    mu = (0.1 * torch.randn((K, D), device=dev, generator=gen)).requires_grad_()
    sg = torch.ones((K, D), device=dev).requires_grad_()
    go = torch.randn((g["n"], K, F), device=dev, generator=gen)

    def step():
        x.grad = ps.grad = mu.grad = sg.grad = None
        GmmConvFuse(G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"], G["permute"], x, ps, mu, sg).backward(go)
    t = timeit(step, iters=50)
    print(json.dumps({"op": "gmmconv fwd+bwd (PubMed-shaped)", "n": g["n"], "nnz": g["nnz"], "K": K, "D": D, "F": F,
                      "ms": round(t, 4), "edges_per_s": round(g["nnz"] / t * 1e3)}), flush=True)


def reddit():
    from dgnn_amd.native import fused_gatconv as fg, mhsddmm, spmm
    g = bench.build_graph("reddit", dev)
    n, nnz = g["n"], g["nnz"]
    H, F = 4, 64
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    x = torch.rand((n, H, F), device=dev, generator=gen) - 0.4
    ar = torch.rand((n, H), device=dev, generator=gen) - 0.6
    ac = torch.rand((n, H), device=dev, generator=gen) - 0.4
    go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5
    val = torch.rand((nnz, H), device=dev, generator=gen)
    out, emax, esum, emask = fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0)
    tf = timeit(lambda: fg.gat_forward(ar, ac, g["row_ptr"], g["col_ind"], 0.2, x, 0.0), iters=5, warm=2)
    tb = timeit(lambda: fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"],
                                        emax, esum, emask, x, ar, ac, go, out_feat=out), iters=5, warm=2)
    ab = bench.algorithmic_bytes(n, nnz, H, F)
    print(json.dumps({"op": "gat fwd+bwd (Reddit-shaped R-MAT)", "n": n, "nnz": nnz, "fwd_ms": round(tf, 3),
                      "bwd_ms": round(tb, 3), "edges_per_s": round(nnz / (tf + tb) * 1e3),
                      "fwd_GBps": round(ab["fwd"] / tf / 1e6, 1), "bwd_GBps": round(ab["bwd_csc"] / tb / 1e6, 1)}), flush=True)
    ts = timeit(lambda: spmm.spmm_csr(g["row_ptr"], g["col_ind"], val, x), iters=5, warm=2)
    td = timeit(lambda: mhsddmm.mhsddmm(g["row_ptr"], g["col_ind"], go, x), iters=int(os.environ.get("DGNN_BENCH_ITERS", "3")), warm=1)
    print(json.dumps({"op": "spmm_csr (Reddit-shaped)", "ms": round(ts, 3),
                      "GBps": round((nnz * (4 * H * F + 4 * H + 4) + n * 4 * H * F) / ts / 1e6, 1)}), flush=True)
    print(json.dumps({"op": "mhsddmm (Reddit-shaped)", "ms": round(td, 3),
                      "GBps": round((nnz * (4 * H * F + 4 * H + 4) + n * 4 * H * F) / td / 1e6, 1)}), flush=True)


def odd_widths(workload="products"):
    """Feature widths that are not multiples of four: the autograd operator (rows padded to the next
    multiple of four, dgnn_amd/operators/fused_gatconv.py::_pad4) against the native module called
    on the unpadded rows; fwd+bwd per call."""
    from dgnn_amd.native import fused_gatconv as fg
    from dgnn_amd.operators import GATConvFuse
    g = bench.build_graph(workload, dev)
    n = g["n"]
    gen = torch.Generator(device=dev); gen.manual_seed(0)
    for H, F in ((1, 47), (4, 41), (1, 7), (3, 6), (2, 30)):
        x = (torch.rand((n, H, F), device=dev, generator=gen) - 0.4).requires_grad_()
        ar = (torch.rand((n, H), device=dev, generator=gen) - 0.6).requires_grad_()
        ac = (torch.rand((n, H), device=dev, generator=gen) - 0.4).requires_grad_()
        go = torch.rand((n, H, F), device=dev, generator=gen) - 0.5

        def op():
            x.grad = ar.grad = ac.grad = None
            GATConvFuse(ar, ac, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], 0.2, x, 0.0).backward(go)

        def native():
            xd, ard, acd = x.detach(), ar.detach(), ac.detach()
            out, emax, esum, emask = fg.gat_forward(ard, acd, g["row_ptr"], g["col_ind"], 0.2, xd, 0.0)
            fg.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"], g["permute"], emax, esum,
                            emask, xd, ard, acd, go, out_feat=out)
        print(json.dumps({"op": f"gat {workload} fwd+bwd", "H": H, "F": F, "operator_padded_ms": round(timeit(op, 3, 1), 3),
                          "native_unpadded_ms": round(timeit(native, 3, 1), 3)}), flush=True)
        del x, go


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("which", nargs="*", default=["edgeconv", "gmm", "reddit"])
    a = ap.parse_args()
    for w in a.which:
        {"edgeconv": edgeconv, "gmm": gmm_pubmed, "reddit": reddit, "knn": knn, "layer": layer, "shapes": shapes,
         "shapes_reddit": lambda: shapes("reddit"), "odd": odd_widths}[w]()
