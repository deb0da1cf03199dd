"""ctypes front-end of the GAT layer's single-pass steps around the fused operator
(csrc/layer.hip): attention score terms (dgNN/layers/gatconv_layer.py:74-75) and the
residual + bias + activation tail (gatconv_layer.py:82-90).  Not a reference module: the
reference runs these steps as chains of elementwise torch kernels."""
import torch

from dgnn_amd import _lib as L

ACT_NONE, ACT_RELU, ACT_ELU, ACT_LEAKY_RELU = 0, 1, 2, 3
_UNSUPPORTED = -2


def attn_scores(feat, attn_l, attn_r):
    """feat [n,h,f], attn_l/attn_r [h,f] (any leading 1) -> (attn_row, attn_col) [n,h], or None
    when the shape is outside the kernel's range (caller keeps the torch expression)."""
    L.f32(feat, "feat", 3), L.f32(attn_l, "attn_l"), L.f32(attn_r, "attn_r")
    n, h, f = feat.shape
    if attn_l.numel() != h * f or attn_r.numel() != h * f:
        raise L.DgnnError("attn_l / attn_r must hold h*f values")
    with torch.cuda.device(feat.device):
        attn_row = torch.empty((n, h), dtype=torch.float32, device=feat.device)
        attn_col = torch.empty((n, h), dtype=torch.float32, device=feat.device)
        rc = L.lib().dgnn_gat_attn_scores(n, h, f, L.p(feat), L.p(attn_l), L.p(attn_r), L.p(attn_row),
                                          L.p(attn_col), L.stream_of(feat))
    if rc == _UNSUPPORTED:
        return None
    L.check(rc, "dgnn_gat_attn_scores")
    return attn_row, attn_col


def attn_scores_backward(feat, attn_l, attn_r, g_row, g_col, grad_feat):
    """Adds the score terms' gradient into ``grad_feat`` IN PLACE and returns
    (grad_attn_l, grad_attn_r) shaped like attn_l / attn_r."""
    L.f32(feat, "feat", 3), L.f32(g_row, "g_row", 2), L.f32(g_col, "g_col", 2), L.f32(grad_feat, "grad_feat", 3)
    n, h, f = feat.shape
    with torch.cuda.device(feat.device):
        gl = torch.empty_like(attn_l)
        gr = torch.empty_like(attn_r)
        nb = L.lib().dgnn_gat_attn_scores_backward_workspace_bytes(n, h, f)
        ws = L.workspace(nb, feat)
        rc = L.lib().dgnn_gat_attn_scores_backward(n, h, f, L.p(feat), L.p(attn_l), L.p(attn_r), L.p(g_row),
                                                   L.p(g_col), L.p(grad_feat), L.p(gl), L.p(gr), L.p(ws), nb,
                                                   L.stream_of(feat))
    L.check(rc, "dgnn_gat_attn_scores_backward")
    return gl, gr


def epilogue(rst, residual, bias, activation, act_param):
    L.f32(rst, "rst")
    hf = rst[0].numel() if rst.size(0) else max(rst.numel(), 1)
    n = rst.size(0)
    if residual is not None:
        L.f32(residual, "residual")
        if residual.numel() != rst.numel():
            raise L.DgnnError("residual must have rst's size")
    if bias is not None:
        L.f32(bias, "bias")
        if bias.numel() != hf:
            raise L.DgnnError("bias must hold one value per (head, feature)")
    with torch.cuda.device(rst.device):
        out = torch.empty_like(rst)
        rc = L.lib().dgnn_gat_epilogue(n, hf, int(activation), float(act_param), L.p(rst), L.p(residual),
                                       L.p(bias), L.p(out), L.stream_of(rst))
    L.check(rc, "dgnn_gat_epilogue")
    return out


def epilogue_backward(out, grad, activation, act_param, want_bias):
    """-> (grad_pre, grad_bias or None)."""
    L.f32(out, "out"), L.f32(grad, "grad")
    n = out.size(0)
    hf = out[0].numel() if n else 1
    with torch.cuda.device(out.device):
        grad_pre = torch.empty_like(grad)
        grad_bias = torch.empty(hf, dtype=torch.float32, device=out.device) if want_bias else None
        nb = L.lib().dgnn_gat_epilogue_backward_workspace_bytes(n, hf) if want_bias else 0
        ws = L.workspace(nb, out)
        rc = L.lib().dgnn_gat_epilogue_backward(n, hf, int(activation), float(act_param), L.p(out), L.p(grad),
                                                L.p(grad_pre), L.p(grad_bias), L.p(ws), nb, L.stream_of(out))
    L.check(rc, "dgnn_gat_epilogue_backward")
    return grad_pre, grad_bias
