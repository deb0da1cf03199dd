#!/usr/bin/env python3
"""Generates the committed golden vectors (tests/golden/*.npz) from the CPU oracle.

The reference itself ships no fixtures and cannot run here (CUDA only), so these vectors are
produced by the oracle's FLOAT64 restatement (pinned by tests/test_oracle.py against the DGL/PyG
formulation and torch autograd) on small seeded inputs; index outputs come from scipy exactly as
the reference's scripts build them.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402
from util import degree_graph, gat_inputs, powerlaw_graph, tiny_graph  # noqa: E402

IDX = ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")


def gat_case(name, g, h, f, p, seed):
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed)
    mask = np.random.default_rng(seed + 1).random((g["nnz"], h), dtype=np.float32) if p > 0 else np.zeros((0, h), np.float32)
    m = mask if p > 0 else None
    out, emax, esum = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, p, m)
    _, emax32, _ = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, p, m)
    gf, gr, gc = oracle.gat_backward_f64(0.2, p, g["row_ptr"], g["col_ind"], m, in_feat, attn_row, attn_col, grad)
    np.savez_compressed(os.path.join(HERE, f"gat_{name}.npz"), n=g["n"], h=h, f=f, attn_drop=p, slope=0.2,
                        in_feat=in_feat, attn_row=attn_row, attn_col=attn_col, grad=grad, edge_mask=mask,
                        out_feat=out, edge_max=emax32, edge_sum=esum, grad_feat=gf, grad_attn_row=gr,
                        grad_attn_col=gc, **{k: g[k] for k in IDX})


def gmm_case(name, g, k, d, f, seed):
    rng = np.random.default_rng(seed)
    node_feat = rng.standard_normal((g["n"], k, f)).astype(np.float32)
    pseudo = rng.random((g["nnz"], d), dtype=np.float32)
    mu = (0.1 * rng.standard_normal((k, d))).astype(np.float32)
    inv_sigma = (1 + 0.1 * rng.standard_normal((k, d))).astype(np.float32)
    gg = rng.standard_normal((g["nnz"], k)).astype(np.float32)
    out = oracle.gmmconv_f64(g["row_ptr"], g["col_ind"], node_feat, pseudo, mu, inv_sigma)
    po, mo, so = oracle.gaussian_bp_f64(pseudo, mu, inv_sigma, gg)
    np.savez_compressed(os.path.join(HERE, f"gmm_{name}.npz"), node_feat=node_feat, pseudo=pseudo, mu=mu,
                        inv_sigma=inv_sigma, grad_gauss=gg, out=out, gauss=oracle.gaussian(pseudo, mu, inv_sigma),
                        pseudo_out=po, mu_out=mo, sigma_out=so, **{k2: g[k2] for k2 in IDX})


def edge_case(name, m, f, k, seed):
    rng = np.random.default_rng(seed)
    src = rng.integers(0, m, (m, k)).astype(np.int32).reshape(-1)
    h_src = rng.integers(-2, 3, (m, f)).astype(np.float32)      # quantised: exact ties
    h_dst = rng.integers(-2, 3, (m, f)).astype(np.float32)
    go = rng.standard_normal((m, f)).astype(np.float32)
    out, idx = oracle.edgeconv_forward(k, src, h_src, h_dst)
    gs = oracle.edgeconv_backward_f64(go, idx.astype(np.int32))
    np.savez_compressed(os.path.join(HERE, f"edgeconv_{name}.npz"), k=k, src_ind=src, h_src=h_src, h_dst=h_dst,
                        grad_out=go, out_feat=out, max_idx=idx.astype(np.int32), grad_src=gs)


def sddmm_spmm_case(name, g, h, f, seed):
    """SDDMM (mhsddmm_kernel.cu:5-81), the permuted copy (mhtranspose_kernel.cu:6-14) and the
    GE-SpMM aggregation (phase 3 of fused_forward_kernel) on one graph."""
    rng = np.random.default_rng(seed)
    a = rng.standard_normal((g["n"], h, f)).astype(np.float32)
    b = rng.standard_normal((g["n"], h, f)).astype(np.float32)
    val = rng.standard_normal((g["nnz"], h)).astype(np.float32)
    np.savez_compressed(os.path.join(HERE, f"sddmm_{name}.npz"), grad=a, feature=b, edge_val=val,
                        sddmm=oracle.mhsddmm_f64(g["row_ptr"], g["col_ind"], a, b),
                        transposed=oracle.mhtranspose(g["permute"], val),
                        spmm=oracle.spmm_csr_f64(g["row_ptr"], g["col_ind"], val, b), **{k: g[k] for k in IDX})


def index_case(name, n, n_edges, seed):
    """COO -> CSR by destination -> CSC + permute with scipy ONLY, the way the reference's scripts
    build them (train_gatconv.py:78-94, test_convert.py:29-30): independent of the oracle."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    key = np.unique(rng.integers(0, n, n_edges).astype(np.int64) * n + rng.integers(0, n, n_edges))
    dst, src = (key // n).astype(np.int32), (key % n).astype(np.int32)       # sorted by (dst, src)
    nnz = key.size
    csr = sp.csr_matrix((np.arange(1, nnz + 1, dtype=np.float64), (dst, src)), shape=(n, n))
    csc = csr.tocsc()
    np.savez_compressed(os.path.join(HERE, f"index_{name}.npz"), n=n, coo_row=dst, coo_col=src,
                        row_ptr=csr.indptr.astype(np.int32), col_ind=csr.indices.astype(np.int32),
                        col_ptr=csc.indptr.astype(np.int32), row_ind=csc.indices.astype(np.int32),
                        permute=(csc.data - 1).astype(np.int32))


def knn_case(name, batch, points, dim, k, seed, quantised):
    """k-NN lists by brute force in float64 where that is unambiguous: quantised coordinates make
    every squared distance an exact small integer in float32 AND float64 (ties -> lower id,
    itself first); the random case is kept only if no two float64 distances of a query differ by
    less than 1e-6 relative (so float32 rounding cannot reorder them)."""
    rng = np.random.default_rng(seed)
    while True:
        x = (rng.integers(-3, 4, (batch, points, dim)).astype(np.float32) if quantised
             else rng.standard_normal((batch, points, dim)).astype(np.float32))
        out = np.empty((batch, points, k), np.int32)
        ok = True
        for b in range(batch):
            d = ((x[b].astype(np.float64)[:, None, :] - x[b].astype(np.float64)[None, :, :]) ** 2).sum(-1)
            order = np.lexsort((np.arange(points)[None, :].repeat(points, 0), d), axis=1)   # by d, then id
            out[b] = order[:, :k] + b * points
            if not quantised:
                ds = np.sort(d, axis=1)[:, :k + 1]
                ok &= bool(np.all(np.diff(ds, axis=1)[:, 1:] > 1e-6 * ds[:, 2:]))
        if ok:
            break
    np.savez_compressed(os.path.join(HERE, f"knn_{name}.npz"), x=x, k=k, src_ind=out.reshape(-1))


def layer_grad_pattern(shape):
    """Deterministic upstream gradient of the layer fixture (not stored: 2.8 MB of noise)."""
    i = np.arange(int(np.prod(shape)), dtype=np.float64)
    return (0.5 * np.sin(0.37 * i + 1.3)).astype(np.float32).reshape(shape)


def layer_case(name, n, n_edges, in_feats, heads, out_feats, seed):
    """One forward + backward of the GATConv LAYER (dgNN/layers/gatconv_layer.py:69-92: projection,
    score terms, fused operator, bias, ELU) at the Cora shape: parameters + input -> output and
    every gradient, from the float64 restatement tests/layer_ref.py.  Stored as float32 (6e-8
    relative: far inside the 1e-5 band they are compared at)."""
    import layer_ref
    from util import random_graph
    g = random_graph(n, n_edges, seed)
    rng = np.random.default_rng(seed)
    hf = heads * out_feats
    params = {"W": (rng.standard_normal((in_feats, hf)) * np.sqrt(2.0 * 2 / (in_feats + hf))).astype(np.float32),
              "attn_l": (rng.standard_normal((1, heads, out_feats)) * np.sqrt(2.0 * 2 / (heads * out_feats + out_feats))).astype(np.float32),
              "attn_r": (rng.standard_normal((1, heads, out_feats)) * np.sqrt(2.0 * 2 / (heads * out_feats + out_feats))).astype(np.float32),
              "bias": (0.1 * rng.standard_normal(hf)).astype(np.float32)}
    feat = rng.standard_normal((n, in_feats)).astype(np.float32)
    grad = layer_grad_pattern((n, heads, out_feats))
    vals, _ = layer_ref.gat_layer_f64(params, g, feat, grad, heads, out_feats, activation="elu")
    np.savez_compressed(os.path.join(HERE, f"layer_{name}.npz"), heads=heads, out_feats=out_feats, feat=feat,
                        **{"param_" + k: v for k, v in params.items()},
                        out=vals["out"].astype(np.float32), grad_feat=vals["feat"].astype(np.float32),
                        **{"grad_" + k: vals[k].astype(np.float32) for k in params},
                        **{k: g[k] for k in IDX})


if __name__ == "__main__":
    if sys.argv[1:] == ["layer"]:
        layer_case("gat_cora", 2708, 10556, 64, 4, 64, 21)
        sys.exit(0)
    gat_case("hub_h1f64", powerlaw_graph(256, 8, hub_degree=200), 1, 64, 0.0, 8)
    gat_case("deg64_h1f100", degree_graph(80, [64, 65, 32, 33, 0, 1] + [3] * 74, 9), 1, 100, 0.0, 9)
    sddmm_spmm_case("deg_h3f6", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 3, 6, 10)
    sddmm_spmm_case("hub_h4f16", powerlaw_graph(128, 11, hub_degree=100), 4, 16, 11)
    index_case("n300", 300, 4000, 12)
    index_case("n50_dense", 50, 3000, 13)
    knn_case("ties_d3_k20", 2, 96, 3, 20, 14, True)
    knn_case("random_d3_k20", 2, 128, 3, 20, 15, False)
    knn_case("random_d64_k8", 1, 100, 64, 8, 16, False)
    gat_case("tiny_h2f8", tiny_graph(), 2, 8, 0.0, 1)
    gat_case("deg_h4f16", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 4, 16, 0.0, 2)
    gat_case("deg_h4f16_drop", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 4, 16, 0.5, 3)
    gat_case("deg_h3f6", degree_graph(24, [3, 0, 7, 24, 1] + [2] * 19, 4), 3, 6, 0.0, 4)
    gmm_case("deg_k3d2f8", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 3, 2, 8, 5)
    edge_case("k20", 48, 12, 20, 6)
    edge_case("k40", 48, 5, 40, 7)
    layer_case("gat_cora", 2708, 10556, 64, 4, 64, 21)
    print(sorted(os.listdir(HERE)))
