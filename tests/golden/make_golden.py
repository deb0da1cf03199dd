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
from util import degree_graph, gat_inputs, tiny_graph  # noqa: E402

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


if __name__ == "__main__":
    gat_case("tiny_h2f8", tiny_graph(), 2, 8, 0.0, 1)
    gat_case("deg_h4f16", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 4, 16, 0.0, 2)
    gat_case("deg_h4f16_drop", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 4, 16, 0.5, 3)
    gat_case("deg_h3f6", degree_graph(24, [3, 0, 7, 24, 1] + [2] * 19, 4), 3, 6, 0.0, 4)
    gmm_case("deg_k3d2f8", degree_graph(40, [0, 1, 31, 32, 33, 40, 5, 0] + [2] * 32, 2), 3, 2, 8, 5)
    edge_case("k20", 48, 12, 20, 6)
    edge_case("k40", 48, 5, 40, 7)
    print(sorted(os.listdir(HERE)))
