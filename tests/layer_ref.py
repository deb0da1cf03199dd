"""Float64 restatements of the reference LAYERS (dgNN/layers/gatconv_layer.py:69-92,
gmmconv_layer.py:56-65) with torch autograd on the CPU, plus first-order summand masses of every
output and gradient, for the layer-level parity tests and the layer golden fixture.

Test infrastructure only (like oracle/): message passing is written with index ops, so nothing
here shares code with the product.  The masses follow util.assert_close's argument: a float32
reduction cannot be closer to the exact value than a few ulp of sum|terms|; along a CHAIN of
reductions the inputs of a stage carry the error of the stage before, so |x| is inflated by the
mass of x when it enters the next stage."""
import numpy as np
import torch

from util import gat_masses_torch, gaussian_bp_masses

F64 = torch.float64


def _t(a, grad=False):
    return torch.tensor(np.asarray(a), dtype=F64, requires_grad=grad)


def _rows(row_ptr):
    rp = torch.as_tensor(np.asarray(row_ptr)).long()
    return torch.repeat_interleave(torch.arange(rp.numel() - 1), rp[1:] - rp[:-1])


def _act(x, activation):
    if activation is None:
        return x
    return {"relu": torch.relu, "elu": torch.nn.functional.elu, "tanh": torch.tanh}[activation](x)


def gat_layer_f64(params, g, feat, grad_out, heads, out_feats, slope=0.2, activation=None, residual=None):
    """GATConv.forward + backward in float64.  params: dict of float32 arrays W [in, H*F],
    attn_l / attn_r [1,H,F], bias [H*F] or None, res_fc.weight [H*F, in] (residual == "linear").
    residual: None | "identity" | "linear" (the Linear case applies res_fc to the layer input, as
    dgnn_amd.layers.GATConv documents).  -> (values, masses): dicts with keys out, feat (its
    gradient) and one entry per parameter gradient."""
    H, F_ = heads, out_feats
    row, col = _rows(g["row_ptr"]), torch.as_tensor(np.asarray(g["col_ind"])).long()
    n = int(np.asarray(g["row_ptr"]).shape[0] - 1)
    X = _t(feat, True)
    P = {k: _t(v, True) for k, v in params.items() if v is not None}
    h = (X @ P["W"]).view(n, H, F_)
    h.retain_grad()
    ar = (P["attn_l"] * h).sum(-1)
    ac = (P["attn_r"] * h).sum(-1)
    e = torch.nn.functional.leaky_relu(ar[row] + ac[col], slope)
    emax = torch.full((n, H), -1e300, dtype=F64).scatter_reduce(0, row[:, None].expand(-1, H), e.detach(), "amax",
                                                                include_self=True)
    ex = torch.exp(e - emax[row])
    a = ex / torch.zeros((n, H), dtype=F64).index_add(0, row, ex)[row]
    rst = torch.zeros((n, H, F_), dtype=F64).index_add(0, row, a[:, :, None] * h[col])
    rst.retain_grad()
    pre = rst
    if residual == "identity":
        pre = pre + h
    elif residual == "linear":
        pre = pre + (X @ P["res_fc.weight"].t()).view(n, H, F_)
    if "bias" in P:
        pre = pre + P["bias"].view(1, H, F_)
    pre.retain_grad()
    out = _act(pre, activation)
    out.backward(_t(grad_out))
    vals = {"out": out.detach().numpy(), "feat": X.grad.numpy()}
    vals.update({k: v.grad.numpy() for k, v in P.items()})

    # ---- masses (first order) ------------------------------------------------------------
    with torch.no_grad():
        aX, aW = X.abs(), P["W"].abs()
        m_h = (aX @ aW).view(n, H, F_)                               # h = feat @ W
        h_in = h.abs() + m_h                                         # |h| as the next stage sees it
        m_ar = (P["attn_l"].abs() * h_in).sum(-1)                    # score terms
        m_ac = (P["attn_r"].abs() * h_in).sum(-1)
        g_pre = pre.grad                                             # d out / d pre (elementwise from grad_out)
        g_rst = rst.grad
        rp_t = torch.as_tensor(np.asarray(g["row_ptr"]))
        m_out, m_gf, m_gr, m_gc = (torch.from_numpy(t) for t in
                                   gat_masses_torch(rp_t, col, ar, ac, h_in, g_rst, slope))
        # a relative error delta in a softmax weight moves the aggregate by |a h| delta; delta is a
        # few ulp of the score masses (exp of a float32 score)
        score_m = (m_ar + torch.zeros((n, H), dtype=F64).index_add(0, row, a * m_ac[col]))
        m_out = m_out * (1.0 + 2.0 * score_m[:, :, None])
        m_pre = m_out + (h_in if residual == "identity" else 0.0)
        if residual == "linear":
            m_pre = m_pre + (aX @ P["res_fc.weight"].abs().t()).view(n, H, F_)
        masses = {"out": m_pre.numpy()}
        # backward chain: grad_h = GAT backward + score-term backward (+ identity residual)
        m_gh = m_gf * (1.0 + 2.0 * score_m.max()) + m_gr[:, :, None] * P["attn_l"].abs() + \
            m_gc[:, :, None] * P["attn_r"].abs()
        gh_in = h.grad.abs() + m_gh
        masses["W"] = (aX.t() @ gh_in.view(n, -1)).numpy()
        m_feat = gh_in.view(n, -1) @ aW.t()
        if residual == "linear":
            m_feat = m_feat + g_pre.abs().view(n, -1) @ P["res_fc.weight"].abs()
            masses["res_fc.weight"] = (g_pre.abs().view(n, -1).t() @ aX).numpy()
        masses["feat"] = m_feat.numpy()
        # d attn_l = sum_n g_row h (g_row = grad_attn_row of the operator), d attn_r likewise
        a_grow, a_gcol = _abs_grad_scores(ar, row, col, a, e, g_rst, h, slope)
        masses["attn_l"] = ((m_gr + a_grow)[:, :, None] * h_in).sum(0, keepdim=True).numpy()
        masses["attn_r"] = ((m_gc + a_gcol)[:, :, None] * h_in).sum(0, keepdim=True).numpy()
        if "bias" in P:
            masses["bias"] = g_pre.abs().view(n, -1).sum(0).numpy()
    return vals, masses


def _abs_grad_scores(ar, row, col, a, e, g_rst, h, slope):
    """|grad_attn_row|, |grad_attn_col| of the fused operator (exact values, float64)."""
    n, H = ar.shape
    ge = (g_rst[row] * h[col]).sum(-1)                                # [E,H]
    S = torch.zeros((n, H), dtype=F64).index_add(0, row, a * ge)
    d = a * (ge - S[row]) * torch.where(e < 0, torch.full_like(e, slope), torch.ones_like(e))
    g_row = torch.zeros((n, H), dtype=F64).index_add(0, row, d)
    g_col = torch.zeros((h.shape[0], H), dtype=F64).index_add(0, col, d)
    return g_row.abs(), g_col.abs()


def gmm_layer_f64(params, g, feat, pseudo, grad_out, n_kernels, out_feats, residual=None):
    """GMMConv.forward + backward in float64 (gmmconv_layer.py:56-65).  params: fc.weight
    [K*F, in], mu, inv_sigma [K,D], bias [F] or None, res_fc.weight [F, in] (residual ==
    "linear").  -> (values, masses) with keys out, feat, pseudo and one per parameter."""
    K, F_ = n_kernels, out_feats
    row, col = _rows(g["row_ptr"]), torch.as_tensor(np.asarray(g["col_ind"])).long()
    n = int(np.asarray(g["row_ptr"]).shape[0] - 1)
    X, PS = _t(feat, True), _t(pseudo, True)
    P = {k: _t(v, True) for k, v in params.items() if v is not None}
    x = (X @ P["fc.weight"].t()).view(n, K, F_)
    x.retain_grad()
    gauss = torch.exp(-0.5 * (((PS[:, None, :] - P["mu"][None]) * P["inv_sigma"][None]) ** 2).sum(-1))   # [E,K]
    agg = torch.zeros((n, K, F_), dtype=F64).index_add(0, row, gauss[:, :, None] * x[col])
    rst = agg.sum(1)
    if residual == "identity":
        rst = rst + X
    elif residual == "linear":
        rst = rst + X @ P["res_fc.weight"].t()
    if "bias" in P:
        rst = rst + P["bias"]
    go = _t(grad_out)
    rst.backward(go)
    vals = {"out": rst.detach().numpy(), "feat": X.grad.numpy(), "pseudo": PS.grad.numpy()}
    vals.update({k: v.grad.numpy() for k, v in P.items()})
    with torch.no_grad():
        aX, aW = X.abs(), P["fc.weight"].abs()
        m_x = (aX @ aW.t()).view(n, K, F_)
        x_in = x.abs() + m_x
        gd = gauss.detach()
        m_out = torch.zeros((n, K, F_), dtype=F64).index_add(0, row, gd[:, :, None] * x_in[col]).sum(1)
        if residual == "identity":
            m_out = m_out + aX
        elif residual == "linear":
            m_out = m_out + aX @ P["res_fc.weight"].abs().t()
        ago = go.abs()
        m_gx = torch.zeros((n, K, F_), dtype=F64).index_add(0, col, gd[:, :, None] * ago[row][:, None, :])
        gx_in = x.grad.abs() + m_gx
        m_gg = (ago[row][:, None, :] * x_in[col]).sum(-1)            # grad_gauss[e,k] = <go[row], x[col,k]>
        mp, mm, ms = gaussian_bp_masses(pseudo, params["mu"], params["inv_sigma"], m_gg.numpy())
        masses = {"out": m_out.numpy(), "pseudo": mp, "mu": mm, "inv_sigma": ms,
                  "fc.weight": (gx_in.view(n, -1).t() @ aX).numpy()}
        m_feat = gx_in.view(n, -1) @ aW
        if residual == "identity":
            m_feat = m_feat + ago
        elif residual == "linear":
            m_feat = m_feat + ago @ P["res_fc.weight"].abs()
            masses["res_fc.weight"] = (ago.t() @ aX).numpy()
        masses["feat"] = m_feat.numpy()
        if "bias" in P:
            masses["bias"] = ago.sum(0).numpy()
    return vals, masses
