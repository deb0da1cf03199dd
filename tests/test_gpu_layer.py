"""GPU parity of the GAT layer's single-pass steps (csrc/layer.hip) against the oracle's float64
restatement of dgNN/layers/gatconv_layer.py:74-90, and of the whole layer against the same
layer written with plain torch ops around the fused operator."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import layer_ref
import oracle
from util import assert_close, random_graph, rmat_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def host(t):
    return t.detach().cpu().numpy()


@pytest.mark.parametrize("h,f", [(4, 64), (1, 64), (8, 32), (3, 6), (2, 100), (1, 256), (5, 7), (4, 128), (3, 64)])
@pytest.mark.parametrize("n", [1, 37, 5000])
def test_attn_scores_forward_backward(h, f, n):
    from dgnn_amd.native import gat_layer_ops as lo
    rng = np.random.default_rng(n + h * 7 + f)
    feat = rng.standard_normal((n, h, f)).astype(np.float32)
    al = rng.standard_normal((1, h, f)).astype(np.float32)
    ar = rng.standard_normal((1, h, f)).astype(np.float32)
    g_row = rng.standard_normal((n, h)).astype(np.float32)
    g_col = rng.standard_normal((n, h)).astype(np.float32)
    gf0 = rng.standard_normal((n, h, f)).astype(np.float32)
    row, col = lo.attn_scores(dev(feat), dev(al), dev(ar))
    r64, c64 = oracle.gat_attn_scores_f64(feat, al, ar)
    mr, mc = oracle.gat_attn_scores_f64(np.abs(feat), np.abs(al), np.abs(ar))
    assert_close(host(row), r64, "attn_row", mass=mr)
    assert_close(host(col), c64, "attn_col", mass=mc)
    gfeat = dev(gf0)
    gl, gr = lo.attn_scores_backward(dev(feat), dev(al), dev(ar), dev(g_row), dev(g_col), gfeat)
    dx, dl, dr = oracle.gat_attn_scores_backward_f64(feat, al, ar, g_row, g_col)
    _, ml, mr2 = oracle.gat_attn_scores_backward_f64(np.abs(feat), np.abs(al), np.abs(ar), np.abs(g_row), np.abs(g_col))
    assert_close(host(gfeat), gf0.astype(np.float64) + dx, "grad_feat accumulated in place")
    assert gl.shape == (1, h, f)
    assert_close(host(gl).reshape(h, f), dl, "grad_attn_l", mass=ml)
    assert_close(host(gr).reshape(h, f), dr, "grad_attn_r", mass=mr2)
    # fixed-order sums: a second run gives the same bits
    gfeat2 = dev(gf0)
    gl2, gr2 = lo.attn_scores_backward(dev(feat), dev(al), dev(ar), dev(g_row), dev(g_col), gfeat2)
    assert torch.equal(gl, gl2) and torch.equal(gr, gr2) and torch.equal(gfeat, gfeat2)


def test_attn_scores_out_of_range_shape_is_reported_not_computed():
    from dgnn_amd.native import gat_layer_ops as lo
    x = torch.randn((10, 1, 300), device=DEV)
    a = torch.randn((1, 1, 300), device=DEV)
    assert lo.attn_scores(x, a, a) is None


@pytest.mark.parametrize("act", [0, 1, 2, 3])
@pytest.mark.parametrize("shape", [(700, 4, 64), (33, 3, 7), (1, 1, 5)])
@pytest.mark.parametrize("with_res,with_bias", [(True, True), (False, True), (True, False), (False, False)])
def test_epilogue_forward_backward(act, shape, with_res, with_bias):
    from dgnn_amd.native import gat_layer_ops as lo
    rng = np.random.default_rng(act + shape[0])
    rst = rng.standard_normal(shape).astype(np.float32)
    res = rng.standard_normal(shape).astype(np.float32) if with_res else None
    bias = rng.standard_normal(shape[1] * shape[2]).astype(np.float32) if with_bias else None
    grad = rng.standard_normal(shape).astype(np.float32)
    param = {0: 0.0, 1: 0.0, 2: 1.3, 3: 0.2}[act]
    out = lo.epilogue(dev(rst), dev(res) if with_res else None, dev(bias) if with_bias else None, act, param)
    want = oracle.gat_epilogue_f64(rst, res, bias, act, param)
    assert_close(host(out), want, "epilogue")
    gp, gb = lo.epilogue_backward(out, dev(grad), act, param, with_bias)
    gp64, gb64 = oracle.gat_epilogue_backward_f64(rst, res, bias, act, param, grad)
    # the kernel differentiates at the float32 output (ELU: a*exp(x) = y + a): where y is close to -a the sum
    # cancels, and its float32 rounding is a few ulp of the summands' mass (|y| + a) |grad|
    mass = np.abs(grad) * (np.abs(want) + abs(param)) if act == 2 else None
    assert_close(host(gp), gp64, "grad_pre", mass=mass)
    if with_bias:
        mass = np.abs(gp64).reshape(shape[0], -1).sum(0)
        assert_close(host(gb), gb64, "grad_bias", mass=mass)
    else:
        assert gb is None


def _plain_layer(layer, graph, feat):
    """The reference layer's forward (gatconv_layer.py:68-92) with torch ops around GATConvFuse."""
    from dgnn_amd.operators import GATConvFuse
    h = torch.matmul(feat, layer.W).view(-1, layer.num_heads, layer.out_feats)
    attn_row = (layer.attn_l * h).sum(dim=-1)
    attn_col = (layer.attn_r * h).sum(dim=-1)
    rst = GATConvFuse(attn_row, attn_col, *graph, layer.negative_slope, h, layer.attn_drop)
    if layer.res_fc is not None:
        src = h if isinstance(layer.res_fc, torch.nn.Identity) else feat
        rst = rst + layer.res_fc(src).view(-1, layer.num_heads, layer.out_feats)
    if layer.bias is not None:
        rst = rst + layer.bias.view(-1, layer.num_heads, layer.out_feats)
    if layer.activation:
        rst = layer.activation(rst)
    return rst


@pytest.mark.parametrize("heads,out_feats,residual,activation", [
    (4, 64, False, None), (4, 64, True, F.elu), (2, 16, True, F.relu), (3, 5, False, torch.tanh),
    (1, 300, False, F.elu)])
def test_gat_layer_matches_the_plain_torch_composition(heads, out_feats, residual, activation):
    from dgnn_amd.layers import GATConv
    g = rmat_graph(11, 1500, 20000, 5)
    graph = [dev(g[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    torch.manual_seed(3)
    in_feats = 40
    layer = GATConv(in_feats, out_feats, heads, residual=residual, activation=activation).to(DEV)
    with torch.no_grad():
        layer.bias.normal_()
    assert sorted(layer.state_dict()) == sorted(["W", "attn_l", "attn_r", "bias"] +
                                                 (["res_fc.weight"] if residual else []))
    feat = torch.randn((g["n"], in_feats), device=DEV, requires_grad=True)
    grad = torch.randn((g["n"], heads, out_feats), device=DEV)
    layer.train()
    out = layer(*graph, feat)
    out.backward(grad)
    got = [out.detach()] + [p.grad.clone() for p in layer.parameters()] + [feat.grad.clone()]
    layer.zero_grad()
    feat.grad = None
    ref = _plain_layer(layer, graph, feat)
    ref.backward(grad)
    want = [ref.detach()] + [p.grad.clone() for p in layer.parameters()] + [feat.grad.clone()]
    names = ["out"] + [n for n, _ in layer.named_parameters()] + ["feat"]
    # both ways of writing the layer sit in the parity band around the float64 restatement of the
    # reference layer (tests/layer_ref.py), each with the summand mass of its reduction chain
    act_name = {None: None, F.elu: "elu", F.relu: "relu", torch.tanh: "tanh"}[activation]
    params = {n_: host(p_) for n_, p_ in layer.named_parameters()}
    res_kind = None if not residual else ("identity" if isinstance(layer.res_fc, torch.nn.Identity) else "linear")
    vals, masses = layer_ref.gat_layer_f64(params, g, host(feat), host(grad), heads, out_feats, activation=act_name,
                                           residual=res_kind)
    for nm, a, b in zip(names, got, want):
        ref64 = vals[nm].reshape(a.shape)
        mass = masses[nm].reshape(a.shape)
        assert_close(host(a), ref64, f"GATConv layer: {nm}", mass=mass)
        assert_close(host(b), ref64, f"GATConv layer written with torch ops: {nm}", mass=mass)
    layer.eval()
    with torch.no_grad():
        ev = layer(*graph, feat)
    assert torch.allclose(ev, got[0], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("k,d,out_feats,residual,bias", [(3, 2, 64, False, True), (2, 3, 16, True, True),
                                                       (3, 2, 41, True, False), (1, 1, 8, False, False)])
def test_gmm_layer_against_the_float64_reference(k, d, out_feats, residual, bias):
    """GMMConv layer (dgNN/layers/gmmconv_layer.py:9-65): state_dict keys, initialisation recipe
    (mu ~ N(0, 0.1), inv_sigma = 1, zero bias: :43-51), forward and every gradient against the
    float64 restatement of the reference layer."""
    from dgnn_amd.layers import GMMConv
    g = random_graph(1200, 9000, 3)
    graph = [dev(g[key]) for key in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")]
    torch.manual_seed(5)
    in_feats = 24
    layer = GMMConv(in_feats, out_feats, d, k, residual=residual, bias=bias).to(DEV)
    want_keys = ["mu", "inv_sigma", "fc.weight"] + (["bias"] if bias else []) + (["res_fc.weight"] if residual else [])
    assert sorted(layer.state_dict()) == sorted(want_keys)
    assert layer.fc.weight.shape == (k * out_feats, in_feats) and layer.mu.shape == (k, d)
    assert bool((layer.inv_sigma == 1).all()) and (not bias or bool((layer.bias == 0).all()))
    assert k * d < 4 or 0.03 < float(layer.mu.detach().std()) < 0.3
    # a state_dict with the reference's keys and shapes loads (checkpoint compatibility)
    ref_state = {key: torch.randn_like(v) for key, v in layer.state_dict().items()}
    layer.load_state_dict(ref_state)
    with torch.no_grad():
        layer.inv_sigma.copy_(1 + 0.1 * torch.randn_like(layer.inv_sigma))
        layer.mu.mul_(0.1)
        layer.fc.weight.mul_(0.2)
    feat = torch.randn((g["n"], in_feats), device=DEV, requires_grad=True)
    pseudo = torch.rand((g["nnz"], d), device=DEV, requires_grad=True)
    grad = torch.randn((g["n"], out_feats), device=DEV)
    out = layer(*graph, feat, pseudo)
    assert out.shape == (g["n"], out_feats)
    out.backward(grad)
    params = {n_: host(p_) for n_, p_ in layer.named_parameters()}
    vals, masses = layer_ref.gmm_layer_f64(params, g, host(feat), host(pseudo), host(grad), k, out_feats,
                                           residual="linear" if residual else None)
    assert_close(host(out), vals["out"], "GMMConv layer: out", mass=masses["out"])
    assert_close(host(feat.grad), vals["feat"], "GMMConv layer: feat.grad", mass=masses["feat"])
    assert_close(host(pseudo.grad), vals["pseudo"], "GMMConv layer: pseudo.grad", mass=masses["pseudo"])
    for n_, p_ in layer.named_parameters():
        assert_close(host(p_.grad), vals[n_], f"GMMConv layer: {n_}.grad", mass=masses[n_])
