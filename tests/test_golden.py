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
ids = lambda fs: [os.path.basename(f)[:-4] for f in fs]


def test_fixtures_present():
    assert len(GAT) >= 4 and len(GMM) >= 1 and len(EDGE) >= 2


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
