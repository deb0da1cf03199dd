"""Pins the CPU oracle (oracle/) by formula equivalence.

The reference ships no golden vectors and cannot be built here (CUDA only), so the
oracle is checked against an INDEPENDENT statement of the same operators: the DGL /
PyG message-passing formulation that the reference's own functional tests compare
with (test/func_test/func_gat_dgl.py:46-68, func_gmm_pyg.py:74-89,
func_edgeconv_pyg.py:24-41), written here with torch index ops in float64, and torch
autograd for every gradient (the reference never tests gradients).
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

import oracle
from util import (ATOL, RTOL, LITERAL_TALLY, assert_close, gat_inputs, gat_masses, gaussian_bp_masses, gmm_case,
                  named_graphs, parity_graphs, random_graph, rmat_graph, spmm_case, tiny_graph,
                  GMM_GRAPHS, GMM_SHAPES, GMM_SUM_GRAPHS, GMM_SUM_SHAPES, SPMM_GRAPHS, SPMM_SHAPES)

GRAPHS = named_graphs()


def rows_of(g):
    return np.repeat(np.arange(g["n"]), np.diff(g["row_ptr"]))


def torch_gat(g, attn_row, attn_col, in_feat, slope, p, mask):
    """u_add_v -> leaky_relu -> edge_softmax -> u_mul_e/sum (func_gat_dgl.py:46-60)."""
    n = g["n"]
    row = torch.from_numpy(rows_of(g)).long()
    col = torch.from_numpy(g["col_ind"]).long()
    ar = torch.tensor(attn_row, dtype=torch.float64, requires_grad=True)
    ac = torch.tensor(attn_col, dtype=torch.float64, requires_grad=True)
    x = torch.tensor(in_feat, dtype=torch.float64, requires_grad=True)
    h = ar.shape[1]
    e = torch.nn.functional.leaky_relu(ar[row] + ac[col], slope)
    emax = torch.full((n, h), float(np.float32(-1e38)), dtype=torch.float64).scatter_reduce(
        0, row[:, None].expand(-1, h), e.detach(), "amax", include_self=True)
    ex = torch.exp(e - emax[row])
    s = torch.zeros((n, h), dtype=torch.float64).index_add(0, row, ex)
    a = ex / s[row]
    if mask is not None:
        keep = torch.from_numpy((mask > np.float32(p)).astype(np.float64))
        a = a * keep / (1.0 - p)
    out = torch.zeros_like(x).index_add(0, row, a[:, :, None] * x[col])
    return out, (ar, ac, x), emax, s


@pytest.mark.parametrize("name", list(GRAPHS))
@pytest.mark.parametrize("h,f", [(4, 64), (1, 16), (3, 6)])
@pytest.mark.parametrize("p", [0.0, 0.5])
def test_gat_forward_backward_vs_autograd(name, h, f, p):
    g = GRAPHS[name]
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed=hash((name, h, f)) % 1000)
    rng = np.random.default_rng(5)
    mask = rng.random((g["nnz"], h), dtype=np.float32) if p > 0 else None
    slope = 0.2
    ref_out, (ar, ac, x), ref_max, ref_sum = torch_gat(g, attn_row, attn_col, in_feat, slope, p, mask)
    ref_out.backward(torch.tensor(grad, dtype=torch.float64))

    o64, m64, s64 = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], slope,
                                           in_feat, p, mask)
    np.testing.assert_allclose(o64, ref_out.detach().numpy(), rtol=1e-11, atol=1e-13)
    np.testing.assert_allclose(m64, ref_max.numpy(), rtol=0, atol=0)
    np.testing.assert_allclose(s64, ref_sum.detach().numpy(), rtol=1e-12)

    o32, m32, s32 = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], slope,
                                       in_feat, p, mask)
    assert_close(o32, o64, "out_feat f32")
    assert_close(s32, s64, "edge_sum f32")
    # max of float inputs is exact in either precision, up to the f32 rounding of the leaky product
    assert_close(m32, m64, "edge_max f32", rtol=1e-6, atol=1e-37 * 1e38)

    gf64, gr64, gc64 = oracle.gat_backward_f64(slope, p, g["row_ptr"], g["col_ind"], mask, in_feat,
                                               attn_row, attn_col, grad)
    np.testing.assert_allclose(gf64, x.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(gr64, ar.grad.numpy(), rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gc64, ac.grad.numpy(), rtol=1e-9, atol=1e-12)

    gf32, gr32, gc32 = oracle.gat_backward(slope, p, g["row_ptr"], g["col_ind"], g["col_ptr"],
                                           g["row_ind"], g["permute"], m32, s32, mask, in_feat,
                                           attn_row, attn_col, grad)
    assert_close(gf32, gf64, "grad_feat f32")
    # attention gradients are sums of a*(g-S) terms that cancel: anchor the tolerance on the
    # magnitude of the summands, not of the (possibly tiny) sum.
    _, _, m_gr, m_gc = gat_masses(g, attn_row, attn_col, in_feat, grad, slope, p)
    assert_close(gr32, gr64, "oracle f32: grad_attn_row", mass=m_gr)
    assert_close(gc32, gc64, "oracle f32: grad_attn_col", mass=m_gc)


def test_gat_inference_equals_forward_without_dropout():
    g = GRAPHS["powerlaw256"]
    in_feat, attn_row, attn_col, _ = gat_inputs(g["n"], 4, 64, 3)
    o1 = oracle.gat_inference(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat)
    o2, _, _ = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    assert np.array_equal(o1, o2)


def test_gat_empty_row_semantics():
    """fused_gatconv_kernel.cu:45,66,90: empty row -> edge_max=-1e38, edge_sum=0, out=0."""
    g = tiny_graph()
    in_feat, attn_row, attn_col, _ = gat_inputs(5, 2, 8, 0)
    o, m, s = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    assert np.all(o[3] == 0) and np.all(s[3] == 0) and np.all(m[3] == np.float32(-1e38))


@pytest.mark.parametrize("name", ["tiny", "er64", "powerlaw256"])
def test_spmm_and_sddmm(name):
    g = GRAPHS[name]
    h, f = 3, 20
    rng = np.random.default_rng(1)
    val = rng.standard_normal((g["nnz"], h)).astype(np.float32)
    a = rng.standard_normal((g["n"], h, f)).astype(np.float32)
    b = rng.standard_normal((g["n"], h, f)).astype(np.float32)
    row = rows_of(g)
    col = g["col_ind"]
    ref = np.zeros((g["n"], h, f))
    np.add.at(ref, row, val[:, :, None].astype(np.float64) * b[col].astype(np.float64))
    np.testing.assert_allclose(oracle.spmm_csr_f64(g["row_ptr"], col, val, b), ref, rtol=1e-12, atol=1e-12)
    assert_close(oracle.spmm_csr(g["row_ptr"], col, val, b), ref, "oracle f32: spmm",
                 mass=oracle.spmm_csr_f64(g["row_ptr"], col, np.abs(val), np.abs(b)))
    ref_dd = np.einsum("ehf,ehf->eh", a[row].astype(np.float64), b[col].astype(np.float64))
    np.testing.assert_allclose(oracle.mhsddmm_f64(g["row_ptr"], col, a, b), ref_dd, rtol=1e-12, atol=1e-12)
    assert_close(oracle.mhsddmm(g["row_ptr"], col, a, b), ref_dd, "oracle f32: sddmm",
                 mass=oracle.mhsddmm_f64(g["row_ptr"], col, np.abs(a), np.abs(b)))


@pytest.mark.parametrize("name", list(GRAPHS))
def test_csr2csc_bit_exact_vs_scipy(name):
    g = GRAPHS[name]
    col_ptr, row_ind, permute = oracle.csr2csc(g["row_ptr"], g["col_ind"])
    assert np.array_equal(col_ptr, g["col_ptr"])
    assert np.array_equal(row_ind, g["row_ind"])
    assert np.array_equal(permute, g["permute"])
    # permute really maps CSC order to CSR positions
    assert np.array_equal(g["col_ind"][permute], np.repeat(np.arange(g["n"]), np.diff(col_ptr)))
    coo_row = rows_of(g).astype(np.int32)
    assert np.array_equal(oracle.coo2csr(coo_row, g["n"]), g["row_ptr"])
    x = np.random.default_rng(0).standard_normal((g["nnz"], 3)).astype(np.float32)
    assert np.array_equal(oracle.mhtranspose(permute, x), x[permute])


def torch_gmm(g, node_feat, pseudo, mu, inv_sigma):
    """PyG GMMConv message (func_gmm_pyg.py:74-89) with aggr='add', per kernel (no sum over K)."""
    row = torch.from_numpy(rows_of(g)).long()
    col = torch.from_numpy(g["col_ind"]).long()
    x = torch.tensor(node_feat, dtype=torch.float64, requires_grad=True)
    ps = torch.tensor(pseudo, dtype=torch.float64, requires_grad=True)
    m = torch.tensor(mu, dtype=torch.float64, requires_grad=True)
    sg = torch.tensor(inv_sigma, dtype=torch.float64, requires_grad=True)
    gauss = torch.exp(-0.5 * (((ps[:, None, :] - m[None]) * sg[None]) ** 2).sum(-1))  # [E,K]
    out = torch.zeros_like(x).index_add(0, row, gauss[:, :, None] * x[col])
    return out, gauss, (x, ps, m, sg)


@pytest.mark.parametrize("name", ["tiny", "er64", "powerlaw256"])
def test_gmm_forward_and_gaussian_bp(name):
    g = GRAPHS[name]
    k, d, f = 3, 2, 16
    rng = np.random.default_rng(2)
    node_feat = rng.standard_normal((g["n"], k, f)).astype(np.float32)
    pseudo = rng.random((g["nnz"], d), dtype=np.float32)
    mu = (0.1 * rng.standard_normal((k, d))).astype(np.float32)
    inv_sigma = (1 + 0.1 * rng.standard_normal((k, d))).astype(np.float32)
    out, gauss, (x, ps, m, sg) = torch_gmm(g, node_feat, pseudo, mu, inv_sigma)
    o64 = oracle.gmmconv_f64(g["row_ptr"], g["col_ind"], node_feat, pseudo, mu, inv_sigma)
    np.testing.assert_allclose(o64, out.detach().numpy(), rtol=1e-11, atol=1e-12)
    assert_close(oracle.gmmconv(g["row_ptr"], g["col_ind"], node_feat, pseudo, mu, inv_sigma), o64,
                 "oracle f32: gmm out",
                 mass=oracle.gmmconv_f64(g["row_ptr"], g["col_ind"], np.abs(node_feat), pseudo, mu, inv_sigma))
    assert_close(oracle.gaussian(pseudo, mu, inv_sigma), gauss.detach().numpy(), "gaussian")
    gg = rng.standard_normal((g["nnz"], k)).astype(np.float32)
    gauss.backward(torch.tensor(gg, dtype=torch.float64))
    po, mo, so = oracle.gaussian_bp_f64(pseudo, mu, inv_sigma, gg)
    np.testing.assert_allclose(po, ps.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(mo, m.grad.numpy(), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(so, sg.grad.numpy(), rtol=1e-10, atol=1e-12)
    po32, mo32, so32 = oracle.gaussian_bp(pseudo, mu, inv_sigma, gg)
    m_po, m_mo, m_so = gaussian_bp_masses(pseudo, mu, inv_sigma, np.abs(gg))
    assert_close(po32, po, "oracle f32: pseudo grad", mass=m_po)
    assert_close(mo32, mo, "oracle f32: mu grad", mass=m_mo)
    assert_close(so32, so, "oracle f32: sigma grad", mass=m_so)


def bitrev5(x):
    return int("{:05b}".format(x)[::-1], 2)


@pytest.mark.parametrize("k", [1, 5, 20, 32, 33, 40, 70])
def test_edgeconv_forward_argmax_rule(k):
    """max over neighbours (PyG aggr='max', func_edgeconv_pyg.py:24-41) plus the reference's
    butterfly tie rule restated in closed form: inside a 32-tile the tied lane with the largest
    bit-reversed lane index wins, later tiles win ties (fused_edgeconv_kernel.cu:31-41)."""
    m, f = 37, 9
    rng = np.random.default_rng(k)
    src = rng.integers(0, m, (m, k)).astype(np.int32)
    # quantised values force many exact ties
    h_src = rng.integers(-2, 3, (m, f)).astype(np.float32)
    h_dst = rng.integers(-2, 3, (m, f)).astype(np.float32)
    out, idx = oracle.edgeconv_forward(k, src.reshape(-1), h_src, h_dst)
    vals = h_dst[:, None, :] + h_src[src]              # [m,k,f]
    assert np.array_equal(out, vals.max(1))
    for r in range(m):
        for c in range(f):
            best = None
            for t0 in range(0, k, 32):
                tile = [(vals[r, j, c], bitrev5(j - t0), src[r, j]) for j in range(t0, min(k, t0 + 32))]
                tv, _, tid = max(tile, key=lambda t: (t[0], t[1]))
                if best is None or tv >= best[0]:
                    best = (tv, tid)
            assert idx[r, c] == best[1], (r, c)
    # backward = scatter-add of grad_out to the arg-max source
    go = rng.standard_normal((m, f)).astype(np.float32)
    gs = oracle.edgeconv_backward_f64(go, idx.astype(np.int32))
    ref = np.zeros((m, f))
    np.add.at(ref, (idx.astype(np.int64), np.arange(f)[None, :].repeat(m, 0)), go.astype(np.float64))
    np.testing.assert_allclose(gs, ref, rtol=1e-12, atol=1e-12)
    assert_close(oracle.edgeconv_backward(go, idx.astype(np.int32)), ref, "oracle f32: edgeconv bwd",
                 mass=oracle.edgeconv_backward_f64(np.abs(go), idx.astype(np.int32)))


def test_reference_arithmetic_needs_mass_term_on_hubs():
    """On a graph with hub rows/columns (1000+ entries) the literal float32 restatement of the
    reference kernels itself leaves the plain rel 1e-5 / abs 1e-6 band around the exact value on
    a few cancelling sums, and sits inside it once 4 ulp of the summand mass are allowed.  The
    GPU tests use the same (mass-aware) band."""
    g = rmat_graph(15, 30000, 1_000_000, 3)      # hub rows / columns of several thousand entries
    h, f = 4, 64
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, seed=11)
    oracle.set_threads(8)
    o32, m32, s32 = oracle.gat_forward(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    o64, _, _ = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    gf32, gr32, gc32 = oracle.gat_backward(0.2, 0.0, g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"],
                                           g["permute"], m32, s32, None, in_feat, attn_row, attn_col, grad)
    gf64, gr64, gc64 = oracle.gat_backward_f64(0.2, 0.0, g["row_ptr"], g["col_ind"], None, in_feat,
                                               attn_row, attn_col, grad)
    oracle.set_threads(1)
    m_out, m_gf, m_gr, m_gc = gat_masses(g, attn_row, attn_col, in_feat, grad, 0.2)
    assert_close(o32, o64, "out", mass=m_out)
    assert_close(gf32, gf64, "grad_feat", mass=m_gf)
    assert_close(gr32, gr64, "grad_attn_row", mass=m_gr)
    assert_close(gc32, gc64, "grad_attn_col", mass=m_gc)
    strict_bad = sum(int((np.abs(a - b) > ATOL + RTOL * np.abs(b)).sum())
                     for a, b in ((gf32, gf64), (gr32, gr64), (gc32, gc64)))
    assert strict_bad > 0, "expected the plain band to be too tight for the reference's own float32 sums"


def test_literal_float32_tally_on_the_inputs_of_the_gpu_tests():
    """the literal float32 restatement of the reference's SpMM / SDDMM / GMM
    arithmetic against the float64 oracle on EXACTLY the inputs of the GPU parity tests
    (util.spmm_case / gmm_case over the same graphs and shapes), same plain band, same mass term.
    The worst literal error per label (x the plain tolerance) lands in LITERAL_TALLY under the GPU
    tests' labels -- the terminal summary prints it, and the GPU tests hold HIP's worst to
    max(1, 2 x this) (tests/test_gpu_parity.py, test_zz_hip_error_within_twice_the_reference_arithmetic)."""
    G = parity_graphs()
    for name in SPMM_GRAPHS:
        g = G[name]
        rp, ci = g["row_ptr"], g["col_ind"]
        for h, f in SPMM_SHAPES:
            val, a, b = spmm_case(g, h, f)
            lit = oracle.spmm_csr(rp, ci, val, b)
            assert_close(lit, oracle.spmm_csr_f64(rp, ci, val, b), "spmm", literal=lit,
                         mass=oracle.spmm_csr_f64(rp, ci, np.abs(val), np.abs(b)))
            lit = oracle.mhsddmm(rp, ci, a, b)
            assert_close(lit, oracle.mhsddmm_f64(rp, ci, a, b), "sddmm", literal=lit,
                         mass=oracle.mhsddmm_f64(rp, ci, np.abs(a), np.abs(b)))
    for name in GMM_GRAPHS:
        g = G[name]
        rp, ci, cp, ri = g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"]
        for k, d, f in GMM_SHAPES:
            x, ps, mu, sg, gg, grad = gmm_case(g, k, d, f)
            lit = oracle.gmmconv(rp, ci, x, ps, mu, sg)
            assert_close(lit, oracle.gmmconv_f64(rp, ci, x, ps, mu, sg), "gmm forward", literal=lit,
                         mass=oracle.gmmconv_f64(rp, ci, np.abs(x), ps, mu, sg))
            psc = ps[g["permute"]]
            lit = oracle.gmmconv(cp, ri, grad, psc, mu, sg)
            assert_close(lit, oracle.gmmconv_f64(cp, ri, grad, psc, mu, sg), "fused gmm backward: grad_node_feat",
                         literal=lit, mass=oracle.gmmconv_f64(cp, ri, np.abs(grad), psc, mu, sg))
            lit = oracle.mhsddmm(rp, ci, grad, x)
            assert_close(lit, oracle.mhsddmm_f64(rp, ci, grad, x), "fused gmm backward: grad_gauss", literal=lit,
                         mass=oracle.mhsddmm_f64(rp, ci, np.abs(grad), np.abs(x)))
    for name in GMM_SUM_GRAPHS:
        g = G[name]
        rp, ci, cp, ri = g["row_ptr"], g["col_ind"], g["col_ptr"], g["row_ind"]
        for k, f in GMM_SUM_SHAPES:
            x, ps, mu, sg, _, go = gmm_case(g, k, 2, f, seed=k * 10 + f, grad_per_kernel=False)
            go_k = np.repeat(go[:, None, :], k, 1)
            psc = ps[g["permute"]]
            lit = oracle.gmmconv(rp, ci, x, ps, mu, sg).sum(1, dtype=np.float32)
            assert_close(lit, oracle.gmmconv_f64(rp, ci, x, ps, mu, sg).sum(1), "GmmConvFuseSum: out", literal=lit,
                         mass=oracle.gmmconv_f64(rp, ci, np.abs(x), ps, mu, sg).sum(1))
            lit = oracle.gmmconv(cp, ri, go_k, psc, mu, sg)
            assert_close(lit, oracle.gmmconv_f64(cp, ri, go_k, psc, mu, sg), "GmmConvFuseSum: grad node_feat",
                         literal=lit, mass=oracle.gmmconv_f64(cp, ri, np.abs(go_k), psc, mu, sg))
    assert set(LITERAL_TALLY) >= {"spmm", "sddmm", "gmm forward", "fused gmm backward: grad_node_feat",
                                  "fused gmm backward: grad_gauss", "GmmConvFuseSum: out",
                                  "GmmConvFuseSum: grad node_feat"}
    print("literal float32 worst error / plain tolerance:", {k: round(v, 3) for k, v in sorted(LITERAL_TALLY.items())})


# ----------------------------------------------------------------------------- k-NN graph
def _knn_numpy(x, k):
    """Independent statement: float64 distances, stable argsort (ties -> lower id)."""
    b, p, _ = x.shape
    x64 = x.astype(np.float64)
    out = np.empty((b, p, k), np.int64)
    for c in range(b):
        d = ((x64[c][:, None, :] - x64[c][None, :, :]) ** 2).sum(-1)
        out[c] = np.argsort(d, axis=1, kind="stable")[:, :k] + c * p
    return out.reshape(-1)


@pytest.mark.parametrize("b,p,dim,k", [(3, 50, 3, 5), (2, 64, 16, 20), (1, 33, 70, 33), (4, 20, 1, 1)])
def test_knn_oracle_matches_numpy(b, p, dim, k):
    rng = np.random.default_rng(b * 100 + p)
    x = rng.standard_normal((b, p, dim)).astype(np.float32)
    got = oracle.knn_graph(x, k)
    want = _knn_numpy(x, k)
    # float32 vs float64 distances can only reorder near-equal neighbours: none at these sizes
    assert np.array_equal(got, want)
    assert np.array_equal(got.reshape(b * p, k)[:, 0], np.arange(b * p)), "a point is its own nearest"


def test_knn_oracle_tie_rule_lower_id_first():
    # integer grid: many exactly equal distances
    g = np.stack(np.meshgrid(np.arange(4), np.arange(4), indexing="ij"), -1).reshape(1, 16, 2).astype(np.float32)
    got = oracle.knn_graph(g, 5).reshape(16, 5)
    want = _knn_numpy(g, 5).reshape(16, 5)
    assert np.array_equal(got, want)
    dup = np.zeros((1, 6, 3), np.float32)              # all points identical
    assert np.array_equal(oracle.knn_graph(dup, 4).reshape(6, 4), np.tile(np.arange(4), (6, 1)))


def test_knn_oracle_nan_and_inf_points_name_the_query_itself():
    """Fewer than k finite distances: the remaining slots hold the query's own id (in range)."""
    rng = np.random.default_rng(4)
    x = rng.standard_normal((2, 30, 3)).astype(np.float32)
    x[0, 5] = np.nan
    x[1, :25] = np.inf
    got = oracle.knn_graph(x, 8).reshape(2, 30, 8)
    assert (got[0, 5] == 5).all() and not (np.delete(got[0], 5, 0) == 5).any()
    assert got[1].min() >= 30 and got[1].max() < 60
    for q in range(25, 30):               # finite queries of cloud 1: the 5 finite points, then itself
        assert sorted(got[1, q, :5]) == list(range(55, 60)) and (got[1, q, 5:] == 30 + q).all()
    for q in range(25):                   # inf queries: inf - x = inf or NaN for every candidate
        assert (got[1, q] == 30 + q).all()
