"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py).

CPU: the oracle's float32 restatement of the reference kernels against the vectors (regression
pin of the oracle).  GPU: the HIP path through the C ABI against the same vectors."""
import glob
import os

import numpy as np
import pytest

import oracle
from util import assert_close, gat_masses, gaussian_bp_masses

HERE = os.path.dirname(os.path.abspath(__file__))
GAT = sorted(glob.glob(os.path.join(HERE, "golden", "gat_*.npz")))
GMM = sorted(glob.glob(os.path.join(HERE, "golden", "gmm_*.npz")))
EDGE = sorted(glob.glob(os.path.join(HERE, "golden", "edgeconv_*.npz")))
SDDMM = sorted(glob.glob(os.path.join(HERE, "golden", "sddmm_*.npz")))
INDEX = sorted(glob.glob(os.path.join(HERE, "golden", "index_*.npz")))
KNN = sorted(glob.glob(os.path.join(HERE, "golden", "knn_*.npz")))
LAYER = sorted(glob.glob(os.path.join(HERE, "golden", "layer_gat_*.npz")))
ids = lambda fs: [os.path.basename(f)[:-4] for f in fs]


def _gat_masses(z):
    g = {k: z[k] for k in ("row_ptr", "col_ind")}
    g["n"] = int(z["n"])
    return gat_masses(g, z["attn_row"], z["attn_col"], z["in_feat"], z["grad"], 0.2, float(z["attn_drop"]))


def test_fixtures_present():
    assert len(GAT) >= 6 and len(GMM) >= 1 and len(EDGE) >= 2
    assert len(SDDMM) >= 2 and len(INDEX) >= 2 and len(KNN) >= 3


@pytest.mark.parametrize("path", GAT, ids=ids(GAT))
def test_oracle_f32_gat_vs_golden(path):
    z = np.load(path)
    p = float(z["attn_drop"])
    mask = z["edge_mask"] if p > 0 else None
    out, emax, esum = oracle.gat_forward(z["attn_row"], z["attn_col"], z["row_ptr"], z["col_ind"], 0.2, z["in_feat"], p, mask)
    m_out, m_gf, m_gr, m_gc = _gat_masses(z)
    assert_close(out, z["out_feat"], "golden (oracle f32): out_feat", mass=m_out)
    assert np.array_equal(emax, z["edge_max"])
    assert_close(esum, z["edge_sum"], "golden (oracle f32): edge_sum")
    gf, gr, gc = oracle.gat_backward(0.2, p, z["row_ptr"], z["col_ind"], z["col_ptr"], z["row_ind"], z["permute"],
                                     emax, esum, mask, z["in_feat"], z["attn_row"], z["attn_col"], z["grad"])
    assert_close(gf, z["grad_feat"], "golden (oracle f32): grad_feat", mass=m_gf)
    assert_close(gr, z["grad_attn_row"], "golden (oracle f32): grad_attn_row", mass=m_gr)
    assert_close(gc, z["grad_attn_col"], "golden (oracle f32): grad_attn_col", mass=m_gc)
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
    assert_close(oracle.mhsddmm(z["row_ptr"], z["col_ind"], z["grad"], z["feature"]), z["sddmm"], "golden (oracle f32): sddmm",
                 mass=oracle.mhsddmm_f64(z["row_ptr"], z["col_ind"], np.abs(z["grad"]), np.abs(z["feature"])))
    assert np.array_equal(oracle.mhtranspose(z["permute"], z["edge_val"]), z["transposed"])
    assert np.array_equal(z["transposed"], z["edge_val"][z["permute"]])
    assert_close(oracle.spmm_csr(z["row_ptr"], z["col_ind"], z["edge_val"], z["feature"]), z["spmm"], "golden (oracle f32): spmm",
                 mass=oracle.spmm_csr_f64(z["row_ptr"], z["col_ind"], np.abs(z["edge_val"]), np.abs(z["feature"])))


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


def _layer_inputs(z):
    """Parameters, input and the (formula-made) upstream gradient of a layer fixture."""
    import sys
    sys.path.insert(0, os.path.join(HERE, "golden"))
    from make_golden import layer_grad_pattern
    params = {k[len("param_"):]: z[k] for k in z.files if k.startswith("param_")}
    heads, out_feats = int(z["heads"]), int(z["out_feats"])
    return params, heads, out_feats, layer_grad_pattern((z["feat"].shape[0], heads, out_feats))


@pytest.mark.parametrize("path", LAYER, ids=ids(LAYER))
def test_layer_reference_vs_golden(path):
    """Regression pin of tests/layer_ref.py (the float64 restatement of the reference GATConv
    layer) on the committed Cora-shaped fixture; stored as float32."""
    import layer_ref
    z = np.load(path)
    params, heads, out_feats, grad = _layer_inputs(z)
    g = {k: z[k] for k in ("row_ptr", "col_ind")}
    vals, _ = layer_ref.gat_layer_f64(params, g, z["feat"], grad, heads, out_feats, activation="elu")
    np.testing.assert_allclose(vals["out"], z["out"], rtol=2e-7, atol=1e-7)
    np.testing.assert_allclose(vals["feat"], z["grad_feat"], rtol=2e-7, atol=1e-7)
    for k in params:
        np.testing.assert_allclose(vals[k], z["grad_" + k], rtol=2e-7, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("path", LAYER, ids=ids(LAYER))
def test_hip_gat_layer_vs_golden(path):
    """one forward + backward of the GATConv layer at the Cora shape, parameters +
    input -> output and all gradients, HIP path against the committed vectors; a state_dict with
    the reference's keys loads into the layer."""
    import torch
    import layer_ref
    from dgnn_amd.layers import GATConv
    z = np.load(path)
    params, heads, out_feats, grad = _layer_inputs(z)
    g = {k: z[k] for k in ("row_ptr", "col_ind")}
    graph = [_dev(z[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    layer = GATConv(z["feat"].shape[1], out_feats, heads, activation=torch.nn.functional.elu).to("cuda:0")
    layer.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()})       # the reference's keys
    feat = _dev(z["feat"]).requires_grad_()
    layer.train()
    out = layer(*graph, feat)
    out.backward(_dev(grad))
    _, masses = layer_ref.gat_layer_f64(params, g, z["feat"], grad, heads, out_feats, activation="elu")
    assert_close(out.detach().cpu().numpy(), z["out"], "layer golden (HIP): out", mass=masses["out"])
    assert_close(feat.grad.cpu().numpy(), z["grad_feat"], "layer golden (HIP): feat.grad", mass=masses["feat"])
    for k, p in layer.named_parameters():
        assert_close(p.grad.cpu().numpy(), z["grad_" + k].reshape(p.shape), f"layer golden (HIP): {k}.grad",
                     mass=masses[k].reshape(p.shape))


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
    m_out, m_gf, m_gr, m_gc = _gat_masses(z)
    assert_close(out.cpu().numpy(), z["out_feat"], "golden (HIP): out_feat", mass=m_out)
    assert np.array_equal(emax.cpu().numpy(), z["edge_max"]), "edge_max is exact"
    assert_close(esum.cpu().numpy(), z["edge_sum"], "golden (HIP): edge_sum")
    inf = fg.gat_inference(ar, ac, G["row_ptr"], G["col_ind"], 0.2, x)
    if p == 0:
        assert_close(inf.cpu().numpy(), z["out_feat"], "golden (HIP): inference", mass=m_out)
    for kw in ({"out_feat": out}, {}):
        gf, gr, gc = fg.gat_backward(0.2, p, G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"], G["permute"],
                                     emax, esum, emask, x, ar, ac, grad, **kw)
        assert_close(gf.cpu().numpy(), z["grad_feat"], "golden (HIP): grad_feat", mass=m_gf)
        assert_close(gr.cpu().numpy(), z["grad_attn_row"], "golden (HIP): grad_attn_row", mass=m_gr)
        assert_close(gc.cpu().numpy(), z["grad_attn_col"], "golden (HIP): grad_attn_col", mass=m_gc)
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
    assert_close(out.cpu().numpy(), z["out"], "golden (HIP): gmm out",
                 mass=oracle.gmmconv_f64(z["row_ptr"], z["col_ind"], np.abs(z["node_feat"]), z["pseudo"], z["mu"], z["inv_sigma"]))
    assert_close(gm.Gaussian(_dev(z["pseudo"]), _dev(z["mu"]), _dev(z["inv_sigma"])).cpu().numpy(), z["gauss"], "gauss")
    po, mo, so = gm.GaussianBp(_dev(z["pseudo"]), _dev(z["mu"]), _dev(z["inv_sigma"]), _dev(z["grad_gauss"]))
    m_po, m_mo, m_so = gaussian_bp_masses(z["pseudo"], z["mu"], z["inv_sigma"], np.abs(z["grad_gauss"]))
    assert_close(po.cpu().numpy(), z["pseudo_out"], "golden (HIP): pseudo grad", mass=m_po)
    assert_close(mo.cpu().numpy(), z["mu_out"], "golden (HIP): mu grad", mass=m_mo)
    assert_close(so.cpu().numpy(), z["sigma_out"], "golden (HIP): sigma grad", mass=m_so)


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
    assert_close(gs.cpu().numpy(), z["grad_src"], "golden (HIP): edgeconv grad_src",
                 mass=oracle.edgeconv_backward_f64(np.abs(z["grad_out"]), z["max_idx"]))


@pytest.mark.gpu
@pytest.mark.parametrize("path", SDDMM, ids=ids(SDDMM))
def test_hip_sddmm_spmm_vs_golden(path):
    from dgnn_amd.native import mhsddmm, mhtranspose, spmm
    z = np.load(path)
    rp, ci = _dev(z["row_ptr"]), _dev(z["col_ind"])
    out = mhsddmm.mhsddmm(rp, ci, _dev(z["grad"]), _dev(z["feature"]))
    assert_close(out.cpu().numpy(), z["sddmm"], "golden (HIP): sddmm",
                 mass=oracle.mhsddmm_f64(z["row_ptr"], z["col_ind"], np.abs(z["grad"]), np.abs(z["feature"])))
    tr = mhtranspose.mhtranspose(_dev(z["permute"]), _dev(z["edge_val"]))
    assert np.array_equal(tr.cpu().numpy(), z["transposed"]), "the permuted copy is exact"
    agg = spmm.spmm_csr(rp, ci, _dev(z["edge_val"]), _dev(z["feature"]))
    assert_close(agg.cpu().numpy(), z["spmm"], "golden (HIP): spmm",
                 mass=oracle.spmm_csr_f64(z["row_ptr"], z["col_ind"], np.abs(z["edge_val"]), np.abs(z["feature"])))


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
