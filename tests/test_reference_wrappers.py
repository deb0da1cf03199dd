"""Drop-in check against the reference's OWN Python callers (CPU container only).

The reference's autograd wrappers (dgNN/operators/*.py) import top-level native modules
(`import fused_gatconv as fused_gat`, ...).  Here those module names are bound to stand-ins that
(1) bind every call to the SIGNATURE of this repo's drop-in module of the same name
(dgnn_amd/native/, csrc/torch_binding.cpp) -- so a call the reference makes that our module could not accept fails
-- and (2) compute through the CPU oracle.  The reference wrappers are then driven forward and
backward and compared with the float64 oracle: argument order, return lists and the positions
of the `None` gradients (fused_gatconv.py:34, fused_gmmconv.py:27, fused_edgeconv.py:22) are
pinned by the reference's own code.  Skipped where /root/reference is absent (GPU box)."""
import importlib
import inspect
import os
import sys
import types

import numpy as np
import pytest
import torch

import oracle
from util import assert_close, gat_inputs, gat_masses, named_graphs

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "dgNN")), reason="reference tree not present")

T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
N = lambda t: t.detach().cpu().numpy()


def _bound(native_mod_name, fn_name, impl):
    """Stand-in for native function `fn_name`: accepts exactly what our drop-in accepts."""
    import dgnn_amd  # noqa: F401
    mod = importlib.import_module(f"dgnn_amd.native.{native_mod_name}")
    sig = inspect.signature(getattr(mod, fn_name))

    def f(*args, **kw):
        b = sig.bind(*args, **kw)
        b.apply_defaults()
        return impl(**b.arguments)
    return f


@pytest.fixture()
def ref_ops(monkeypatch):
    def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop, edge_mask, out, neg, saved):
        # the reference's wrapper passes seven arguments: the keywords keep their defaults (neg=False, saved=True: on
        # the GPU the fourth result then carries the forward state, tests/test_gpu_parity.py); the oracle-backed
        # stand-in returns the empty mask of attn_drop == 0
        assert edge_mask is None and out is None and neg is False and saved is True
        out, emax, esum = oracle.gat_forward(N(attn_row), N(attn_col), N(row_ptr), N(col_ind), negative_slope,
                                             N(in_feat), attn_drop, None)
        return [T(out), T(emax), T(esum), torch.empty((0, attn_row.size(1)))]

    def gat_backward(negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                     edge_mask, in_feat, attn_row, attn_col, grad, out_feat, row_dot, out, out_neg=None, a_neg=None):
        gf, gr, gc = oracle.gat_backward(negative_slope, attn_drop, N(row_ptr), N(col_ind), N(col_ptr), N(row_ind),
                                         N(permute), N(edge_max), N(edge_sum), None, N(in_feat), N(attn_row),
                                         N(attn_col), N(grad))
        return [T(gf), T(gr), T(gc)]

    def gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
        return T(oracle.gat_inference(N(attn_row), N(attn_col), N(row_ptr), N(col_ind), negative_slope, N(in_feat)))

    mods = {
        "fused_gatconv": dict(gat_forward=_bound("fused_gatconv", "gat_forward", gat_forward),
                              gat_backward=_bound("fused_gatconv", "gat_backward", gat_backward),
                              gat_inference=_bound("fused_gatconv", "gat_inference", gat_inference)),
        "fused_gmmconv": dict(
            gmmconv=_bound("fused_gmmconv", "gmmconv", lambda csrptr, colind, node_feat, pseudo, mu, inv_sigma: T(
                oracle.gmmconv(N(csrptr), N(colind), N(node_feat), N(pseudo), N(mu), N(inv_sigma)))),
            GaussianBp=_bound("fused_gmmconv", "GaussianBp", lambda pseudo, mu, inv_sigma, grad_out: [
                T(a) for a in oracle.gaussian_bp(N(pseudo), N(mu), N(inv_sigma), N(grad_out))])),
        "mhsddmm": dict(mhsddmm=_bound("mhsddmm", "mhsddmm", lambda rowptr, colind, grad, feature: T(
            oracle.mhsddmm(N(rowptr), N(colind), N(grad), N(feature))))),
        "mhtranspose": dict(mhtranspose=_bound("mhtranspose", "mhtranspose", lambda permute, attention: T(
            oracle.mhtranspose(N(permute), N(attention))))),
        "fused_edgeconv": dict(
            edgeconv_forward=_bound("fused_edgeconv", "edgeconv_forward",
                                    lambda k, src_ind, h_src, h_dst, return_int_idx: [T(a) for a in oracle.edgeconv_forward(
                                        k, N(src_ind), N(h_src), N(h_dst))]),
            edgeconv_backward=_bound("fused_edgeconv", "edgeconv_backward", lambda grad_out, max_idx, src_ind: T(
                oracle.edgeconv_backward(N(grad_out), N(max_idx))))),
    }
    for name, fns in mods.items():
        m = types.ModuleType(name)
        for k, v in fns.items():
            setattr(m, k, v)
        monkeypatch.setitem(sys.modules, name, m)
    monkeypatch.syspath_prepend(os.path.join(REF, "dgNN", "operators"))
    out = {}
    for name in ("fused_gatconv", "fused_gmmconv", "fused_edgeconv"):
        spec = importlib.util.spec_from_file_location(f"_ref_{name}", os.path.join(REF, "dgNN", "operators", f"{name}.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        out[name] = mod
    return out


def test_reference_gat_wrapper_runs_on_our_module_signatures(ref_ops):
    g = named_graphs()["powerlaw256"]
    h, f = 2, 8
    in_feat, attn_row, attn_col, grad = gat_inputs(g["n"], h, f, 1)
    G = {k: T(g[k]) for k in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")}
    x, ar, ac = T(in_feat).requires_grad_(), T(attn_row).requires_grad_(), T(attn_col).requires_grad_()
    out = ref_ops["fused_gatconv"].GATConvFuse(ar, ac, G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"],
                                               G["permute"], 0.2, x, 0.0)
    out.backward(T(grad))
    o64, _, _ = oracle.gat_forward_f64(attn_row, attn_col, g["row_ptr"], g["col_ind"], 0.2, in_feat, 0.0)
    gf64, gr64, gc64 = oracle.gat_backward_f64(0.2, 0.0, g["row_ptr"], g["col_ind"], None, in_feat, attn_row, attn_col, grad)
    # the one tolerance of tests/util.py: rel 1e-5 / abs 1e-6 + 4 ulp of each reduction's summand mass
    m_out, m_gf, m_gr, m_gc = gat_masses(g, attn_row, attn_col, in_feat, grad, 0.2)
    assert_close(N(out), o64, "reference wrapper: out", mass=m_out)
    assert_close(N(x.grad), gf64, "reference wrapper: grad_feat", mass=m_gf)
    assert_close(N(ar.grad), gr64, "reference wrapper: grad_attn_row", mass=m_gr)
    assert_close(N(ac.grad), gc64, "reference wrapper: grad_attn_col", mass=m_gc)
    inf = ref_ops["fused_gatconv"].GATConvFuse_inference(ar.detach(), ac.detach(), G["row_ptr"], G["col_ind"], 0.2, x.detach())
    assert_close(N(inf), o64, "reference wrapper: inference", mass=m_out)


def test_reference_gmm_and_edgeconv_wrappers(ref_ops):
    g = named_graphs()["er64"]
    k, d, f = 3, 2, 8
    rng = np.random.default_rng(0)
    G = {k2: T(g[k2]) for k2 in ("row_ptr", "col_ind", "col_ptr", "row_ind", "permute")}
    x = T(rng.standard_normal((g["n"], k, f)).astype(np.float32)).requires_grad_()
    ps = T(rng.random((g["nnz"], d), dtype=np.float32)).requires_grad_()
    mu = T((0.1 * rng.standard_normal((k, d))).astype(np.float32)).requires_grad_()
    sg = T(np.ones((k, d), np.float32)).requires_grad_()
    out = ref_ops["fused_gmmconv"].GmmConvFuse(G["row_ptr"], G["col_ind"], G["col_ptr"], G["row_ind"], G["permute"], x, ps, mu, sg)
    out.backward(torch.ones_like(out))
    assert all(t.grad is not None and t.grad.shape == t.shape for t in (x, ps, mu, sg))
    assert_close(N(out), oracle.gmmconv_f64(g["row_ptr"], g["col_ind"], N(x), N(ps), N(mu), N(sg)), "reference wrapper: gmm out",
                 mass=oracle.gmmconv_f64(g["row_ptr"], g["col_ind"], np.abs(N(x)), N(ps), N(mu), N(sg)))
    m, kk, ff = 40, 7, 5
    src = T(rng.integers(0, m, m * kk).astype(np.int32))
    hs = T(rng.standard_normal((m, ff)).astype(np.float32)).requires_grad_()
    hd = T(rng.standard_normal((m, ff)).astype(np.float32)).requires_grad_()
    o = ref_ops["fused_edgeconv"].EdgeConvFuse(kk, src, hs, hd)
    o.backward(torch.ones_like(o))
    ref_out, ref_idx = oracle.edgeconv_forward(kk, N(src), N(hs), N(hd))
    assert np.array_equal(N(o), ref_out)
    # sums of ones: exact in float32, no band needed beyond the plain one
    assert_close(N(hs.grad), oracle.edgeconv_backward_f64(np.ones((m, ff), np.float32), ref_idx.astype(np.int32)),
                 "reference wrapper: edgeconv grad_src")
    assert torch.equal(hd.grad, torch.ones_like(hd))
