"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle's float32 restatement of the reference kernels against the vectors (regression
pin of the oracle).  GPU: the HIP path through the C ABI against the same vectors."""
import glob
import os

import numpy as np
import pytest

import oracle
from util import assert_close

HERE = os.path.dirname(os.path.abspath(__file__))
GAT = sorted(glob.glob(os.path.join(HERE, "golden", "gat_*.npz")))
GMM = sorted(glob.glob(os.path.join(HERE, "golden", "gmm_*.npz")))
EDGE = sorted(glob.glob(os.path.join(HERE, "golden", "edgeconv_*.npz")))
SDDMM = sorted(glob.glob(os.path.join(HERE, "golden", "sddmm_*.npz")))
INDEX = sorted(glob.glob(os.path.join(HERE, "golden", "index_*.npz")))
KNN = sorted(glob.glob(os.path.join(HERE, "golden", "knn_*.npz")))
ids = lambda fs: [os.path.basename(f)[:-4] for f in fs]


def test_fixtures_present():
    assert len(GAT) >= 6 and len(GMM) >= 1 and len(EDGE) >= 2
    assert len(SDDMM) >= 2 and len(INDEX) >= 2 and len(KNN) >= 3


@pytest.mark.parametrize("path", GAT, ids=ids(GAT))
def test_oracle_f32_gat_vs_golden(path):
    z = np.load(path)
    p = float(z["attn_drop"])
    mask = z["edge_mask"] if p > 0 else None
    out, emax, esum = oracle.gat_forward(z["attn_row"], z["attn_col"], z["row_ptr"], z["col_ind"], 0.2, z["in_feat"], p, mask)
    assert_close(out, z["out_feat"], "out_feat")
    assert np.array_equal(emax, z["edge_max"])
    assert_close(esum, z["edge_sum"], "edge_sum")
    gf, gr, gc = oracle.gat_backward(0.2, p, z["row_ptr"], z["col_ind"], z["col_ptr"], z["row_ind"], z["permute"],
                                     emax, esum, mask, z["in_feat"], z["attn_row"], z["attn_col"], z["grad"])
    assert_close(gf, z["grad_feat"], "grad_feat")
    assert_close(gr, z["grad_attn_row"], "grad_attn_row", atol=2e-6)
    assert_close(gc, z["grad_attn_col"], "grad_attn_col", atol=2e-6)
    cp, ri, pm = oracle.csr2csc(z["row_ptr"], z["col_ind"])
    assert np.array_equal(cp, z["col_ptr"]) and np.array_equal(ri, z["row_ind"]) and np.array_equal(pm, z["permute"])


@pytest.mark.parametrize("path", EDGE, ids=ids(EDGE))
def test_oracle_edgeconv_vs_golden(path):
    z = np.load(path)
    out, idx = oracle.edgeconv_forward(int(z["k"]), z["src_ind"], z["h_src"], z["h_dst"])
    assert np.array_equal(out, z["out_feat"]) and np.array_equal(idx.astype(np.int32), z["max_idx"])


@pytest.mark.parametrize("path", SDDMM, ids=ids(SDDMM))
def test_oracle_sddmm_spmm_vs_golden(path):
    z = np.load(path)
    assert_close(oracle.mhsddmm(z["row_ptr"], z["col_ind"], z["grad"], z["feature"]), z["sddmm"], "sddmm", atol=1e-5)
    assert np.array_equal(oracle.mhtranspose(z["permute"], z["edge_val"]), z["transposed"])
    assert np.array_equal(z["transposed"], z["edge_val"][z["permute"]])
    assert_close(oracle.spmm_csr(z["row_ptr"], z["col_ind"], z["edge_val"], z["feature"]), z["spmm"], "spmm", atol=1e-5)


@pytest.mark.parametrize("path", INDEX, ids=ids(INDEX))
def test_oracle_index_work_vs_scipy_golden(path):
    """The fixture comes from scipy alone (the reference's own recipe), not from the oracle."""
    z = np.load(path)
    assert np.array_equal(oracle.coo2csr(z["coo_row"], int(z["n"])), z["row_ptr"])
    cp, ri, pm = oracle.csr2csc(z["row_ptr"], z["col_ind"])
    assert np.array_equal(cp, z["col_ptr"]) and np.array_equal(ri, z["row_ind"]) and np.array_equal(pm, z["permute"])


@pytest.mark.parametrize("path", KNN, ids=ids(KNN))
def test_oracle_knn_vs_float64_golden(path):
    """The fixture is a float64 brute force with an unambiguous order, not the oracle's chain."""
    z = np.load(path)
    assert np.array_equal(oracle.knn_graph(z["x"], int(z["k"])), z["src_ind"])


def _dev(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.mark.gpu
@pytest.mark.parametrize("path", GAT, ids=ids(GAT))
def test_hip_gat_vs_golden(path):
    import dgnn_amd
    from dgnn_amd.native import fused_gatconv as fg
    z = np.load(path)
    p = float(z["attn_drop"])
    G = {k: _dev(z[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")}
    mask = _dev(z["edge_mask"]) if p > 0 else None
    x, ar, ac, grad = _dev(z["in_feat"]), _dev(z["attn_row"]), _dev(z["attn_col"]), _dev(z["grad"])
    out, emax, esum, emask = fg.gat_forward(ar, ac, G["row_ptr"], G["col_ind"], 0.2, x, p, edge_mask=mask)
    assert_close(out.cpu().numpy(), z["out_feat"], "out_feat")
    assert np.array_equal(emax.cpu().numpy(), z["edge_max"]), "edge_max is exact"
    assert_close(esum.cpu().numpy(), z["edge_sum"], "edge_sum")
    inf = fg.gat_inference(ar, ac, G["row_ptr"], G["col_ind"], 0.2, x)
    if p == 0:
        assert_close(inf.cpu().numpy(), z["out_feat"], "inference")
    for kw in ({"out_feat": out}, {}):
        gf, gr, gc = fg.gat_backward(0.2, p, G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"], G["permute"],
                                     emax, esum, emask, x, ar, ac, grad, **kw)
        assert_close(gf.cpu().numpy(), z["grad_feat"], "grad_feat")
        assert_close(gr.cpu().numpy(), z["grad_attn_row"], "grad_attn_row", atol=2e-6)
        assert_close(gc.cpu().numpy(), z["grad_attn_col"], "grad_attn_col", atol=2e-6)
    from dgnn_amd.native import format_conversion as fc
    cp, ri, pm = fc.csr2csc_int(G["row_ptr"], G["col_ind"])
    assert np.array_equal(cp.cpu().numpy(), z["col_ptr"]) and np.array_equal(ri.cpu().numpy(), z["row_ind"])
    assert np.array_equal(pm.cpu().numpy(), z["permute"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", GMM, ids=ids(GMM))
def test_hip_gmm_vs_golden(path):
    from dgnn_amd.native import fused_gmmconv as gm
    z = np.load(path)
    out = gm.gmmconv(_dev(z["row_ptr"]), _dev(z["col_ind"]), _dev(z["node_feat"]), _dev(z["pseudo"]), _dev(z["mu"]),
                     _dev(z["inv_sigma"]))
    assert_close(out.cpu().numpy(), z["out"], "gmm out", atol=1e-5)
    assert_close(gm.Gaussian(_dev(z["pseudo"]), _dev(z["mu"]), _dev(z["inv_sigma"])).cpu().numpy(), z["gauss"], "gauss")
    po, mo, so = gm.GaussianBp(_dev(z["pseudo"]), _dev(z["mu"]), _dev(z["inv_sigma"]), _dev(z["grad_gauss"]))
    assert_close(po.cpu().numpy(), z["pseudo_out"], "pseudo grad", atol=1e-5)
    assert_close(mo.cpu().numpy(), z["mu_out"], "mu grad", rtol=1e-4, atol=1e-5)
    assert_close(so.cpu().numpy(), z["sigma_out"], "sigma grad", rtol=1e-4, atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("path", EDGE, ids=ids(EDGE))
def test_hip_edgeconv_vs_golden(path):
    from dgnn_amd.native import fused_edgeconv as fe
    z = np.load(path)
    out, idx_f, idx_i = fe.edgeconv_forward(int(z["k"]), _dev(z["src_ind"]), _dev(z["h_src"]), _dev(z["h_dst"]),
                                            return_int_idx=True)
    assert np.array_equal(out.cpu().numpy(), z["out_feat"])
    assert np.array_equal(idx_i.cpu().numpy(), z["max_idx"]), "arg-max ids are bit-exact"
    gs = fe.edgeconv_backward(_dev(z["grad_out"]), idx_i)
    assert_close(gs.cpu().numpy(), z["grad_src"], "grad_src", atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("path", SDDMM, ids=ids(SDDMM))
def test_hip_sddmm_spmm_vs_golden(path):
    from dgnn_amd.native import mhsddmm, mhtranspose, spmm
    z = np.load(path)
    rp, ci = _dev(z["row_ptr"]), _dev(z["col_ind"])
    out = mhsddmm.mhsddmm(rp, ci, _dev(z["grad"]), _dev(z["feature"]))
    assert_close(out.cpu().numpy(), z["sddmm"], "sddmm", atol=1e-5)
    tr = mhtranspose.mhtranspose(_dev(z["permute"]), _dev(z["edge_val"]))
    assert np.array_equal(tr.cpu().numpy(), z["transposed"]), "the permuted copy is exact"
    agg = spmm.spmm_csr(rp, ci, _dev(z["edge_val"]), _dev(z["feature"]))
    assert_close(agg.cpu().numpy(), z["spmm"], "spmm", atol=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("path", INDEX, ids=ids(INDEX))
def test_hip_index_work_vs_scipy_golden(path):
    import torch
    from dgnn_amd.native import format_conversion as fc
    from dgnn_amd.graph import prepare_graph
    z = np.load(path)
    n = int(z["n"])
    assert np.array_equal(fc.coo2csr(_dev(z["coo_row"]), n).cpu().numpy(), z["row_ptr"])
    cp, ri, pm = fc.csr2csc_int(_dev(z["row_ptr"]), _dev(z["col_ind"]))
    assert np.array_equal(cp.cpu().numpy(), z["col_ptr"]) and np.array_equal(ri.cpu().numpy(), z["row_ind"])
    assert np.array_equal(pm.cpu().numpy(), z["permute"])
    # the reference's float-valued variant (convert.cpp:16-25): exact below 2^24 entries
    cp2, ri2, val = fc.csr2csc(_dev(z["row_ptr"]), _dev(z["col_ind"]),
                               torch.arange(z["col_ind"].size, dtype=torch.float32, device="cuda:0"))
    assert np.array_equal(cp2.cpu().numpy(), z["col_ptr"]) and np.array_equal(val.cpu().numpy(), z["permute"].astype(np.float32))
    # whole recipe from shuffled COO (self loops as they are)
    rng = np.random.default_rng(0)
    order = rng.permutation(z["coo_row"].size)
    names = ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")
    g = dict(zip(names, prepare_graph(_dev(z["coo_col"][order]), _dev(z["coo_row"][order]), n, self_loops="keep")))
    for k in names:
        assert np.array_equal(g[k].cpu().numpy(), z[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", KNN, ids=ids(KNN))
def test_hip_knn_vs_float64_golden(path):
    from dgnn_amd.native import knn
    z = np.load(path)
    got = knn.knn_graph(_dev(z["x"]), int(z["k"]))
    assert np.array_equal(got.cpu().numpy().reshape(-1), z["src_ind"]), "k-NN ids are bit-exact"
