"""Autograd operator of the fused GAT path.  API of dgNN/operators/fused_gatconv.py:6-34."""
import torch

from ..native import fused_gatconv as fused_gat
from ._pad import pad4 as _pad4, pad_to as _pad_to


# Device-resident {seed, counter} per GPU for dropout under HIP-graph capture.  A captured launch
# freezes its scalar arguments: a host-drawn seed would replay ONE mask forever.  Under capture the
# operator therefore copies the pair (device-to-device, captured), advances the counter (captured)
# and hands the COPY to the kernels, forward and backward: every replay draws a fresh mask and
# the backward sees its forward's.  The pair is created by the first eager dropout call on a
# device (or init_capture_rng): it cannot be created inside a capture (host-to-device copy).
_CAPTURE_RNG = {}


def init_capture_rng(device, seed=None):
    """Create (or reseed) the device {seed, counter} pair dropout draws from under HIP-graph
    capture.  Called implicitly by any eager call with attn_drop > 0."""
    if seed is None:
        seed = int(torch.randint(0, 1 << 62, (1,)).item())
    state = torch.tensor([seed, 0], dtype=torch.int64, device=torch.device(device))
    # keyed by the tensor's own device: "cuda" and "cuda:0" are one GPU, and the lookup
    # (in_feat.device) always carries the index
    _CAPTURE_RNG[state.device] = state
    return state


def _forward_with_dropout(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop):
    """-> (out_feat, edge_max, edge_sum, edge_mask or None, seed or None, out_neg, a_neg).  The
    last two (None where the kernels have no such pair) are the aggregate and the softmax-weight sum
    over the negative-score edges: the backward turns them into grad_attn_row per row instead of
    scattering [nnz,h] edge gradients and summing them (include/dgnn_hip.h).  With attn_drop > 0 the
    mask is not drawn and kept ([nnz,h] floats: 1 GB per layer at the products shape, what the
    reference saves, fused_gatconv.py:18) but regenerated in the backward from a 62-bit seed taken
    from torch's CPU generator (reproducible under torch.manual_seed, no device sync); shapes the
    seeded kernels do not take get the SAME mask materialised.  Under HIP-graph capture the seed
    is a device tensor instead (see _CAPTURE_RNG)."""
    if not attn_drop > 0:
        out, emax, esum, mask, on, an = fused_gat.gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope,
                                                              in_feat, attn_drop, neg=True)
        return out, emax, esum, mask, None, on, an
    if torch.cuda.is_current_stream_capturing():
        from .._lib import DgnnError
        state = _CAPTURE_RNG.get(in_feat.device)
        if state is None:
            raise DgnnError("attn_drop > 0 under HIP-graph capture needs the device seed created beforehand: run "
                            "one eager step first (the usual warm-up) or call "
                            "dgnn_amd.operators.fused_gatconv.init_capture_rng(device)")
        mine = state.clone()                      # captured copy: what THIS forward and its backward read
        state[1:].add_(1)                         # captured: the next forward / the next replay draws another mask
        res = fused_gat.gat_forward_seeded_dev(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat,
                                               attn_drop, mine, neg=True)
        if res is None:
            raise DgnnError("attn_drop > 0 under HIP-graph capture: this shape is outside the seeded kernels "
                            "(heads > 8 or misaligned rows); capture without dropout or run it eagerly")
        return res[0], res[1], res[2], None, mine, res[3], res[4]
    seed = int(torch.randint(0, 1 << 62, (1,)).item())
    if in_feat.device not in _CAPTURE_RNG:
        # derived from this call's seed: no extra draw from torch's generator (the sequence of
        # eager seeds under torch.manual_seed stays what it was)
        init_capture_rng(in_feat.device, seed=seed ^ 0x2545F4914F6CDD1D)
    res = fused_gat.gat_forward_seeded(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop, seed,
                                       neg=True)
    if res is not None:
        return res[0], res[1], res[2], None, seed, res[3], res[4]
    mask = fused_gat.dropout_mask(col_ind.size(0), attn_row.size(1), seed, 0, in_feat)
    out, emax, esum, mask = fused_gat.gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat,
                                                  attn_drop, edge_mask=mask)
    return out, emax, esum, mask, None, None, None


def _backward_with_dropout(ctx, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask,
                           in_feat, attn_row, attn_col, grad_out, out_feat, out_neg=None, a_neg=None):
    if isinstance(ctx.seed, torch.Tensor):        # forward ran under capture: device pair
        return fused_gat.gat_backward_seeded_dev(ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr,
                                                 row_ind, permute, edge_max, edge_sum, ctx.seed, in_feat, attn_row,
                                                 attn_col, grad_out, out_feat=out_feat, out_neg=out_neg, a_neg=a_neg)
    if ctx.seed is not None:
        res = fused_gat.gat_backward_seeded(ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr, row_ind,
                                            permute, edge_max, edge_sum, ctx.seed, 0, in_feat, attn_row, attn_col,
                                            grad_out, out_feat=out_feat, out_neg=out_neg, a_neg=a_neg)
        if res is not None:
            return res
        edge_mask = fused_gat.dropout_mask(col_ind.size(0), attn_row.size(1), ctx.seed, 0, in_feat)
    return fused_gat.gat_backward(ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute,
                                  edge_max, edge_sum, edge_mask, in_feat, attn_row, attn_col, grad_out,
                                  out_feat=out_feat, out_neg=out_neg, a_neg=a_neg)


def GATConvFuse(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope,
                in_feat, attn_drop):
    """dgNN/operators/fused_gatconv.py:6-7 (same argument order)."""
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute,
                                  negative_slope, in_feat, attn_drop)


def GATConvFuse_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """dgNN/operators/fused_gatconv.py:9-10."""
    f = in_feat.size(2)
    out = fused_gat.gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, _pad4(in_feat))
    return out if out.size(2) == f else out[..., :f]


class FusedGATFunction(torch.autograd.Function):
    """dgNN/operators/fused_gatconv.py:12-34.  Differences, both invisible to callers: the
    forward output is saved too (it turns the backward's row term into one dense dot product,
    removing a whole gather pass), and no [nnz,h] mask is kept: none exists at attn_drop == 0, and
    at attn_drop > 0 the backward regenerates it from a seed (_forward_with_dropout)."""

    @staticmethod
    def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute,
                negative_slope, in_feat, attn_drop):
        attn_row = attn_row.contiguous()
        attn_col = attn_col.contiguous()
        ctx.f = in_feat.size(2)
        in_feat = _pad4(in_feat.contiguous())          # rows narrower than four floats run padded (see _pad4)
        out_feat, edge_max, edge_sum, edge_mask, ctx.seed, out_neg, a_neg = _forward_with_dropout(
            attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop)
        ctx.save_for_backward(row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                              edge_mask, in_feat, attn_row, attn_col, out_feat, out_neg, a_neg)
        ctx.negative_slope = negative_slope
        ctx.attn_drop = attn_drop
        return out_feat if out_feat.size(2) == ctx.f else out_feat[..., :ctx.f]

    @staticmethod
    def backward(ctx, grad_out):
        (row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat,
         attn_row, attn_col, out_feat, out_neg, a_neg) = ctx.saved_tensors
        grad_out = _pad_to(grad_out, in_feat.size(2)).contiguous()   # fused_gatconv.py:26; the forward's width
        grad_feat, grad_attn_row, grad_attn_col = _backward_with_dropout(
            ctx, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat,
            attn_row, attn_col, grad_out, out_feat, out_neg, a_neg)
        if grad_feat.size(2) != ctx.f:
            grad_feat = grad_feat[..., :ctx.f]
        # same tuple layout as fused_gatconv.py:34
        return grad_attn_row, grad_attn_col, None, None, None, None, None, None, grad_feat, None


# ------------------------------------------------------------------------------------------
# Layer-level fusion (no counterpart in dgNN/operators): the score terms
# computed from the projected features in one pass, the fused operator, and in the backward
# the score terms' gradient accumulated into grad_feat in place.
from ..native import gat_layer_ops as _lo  # noqa: E402


def GATConvFuseFromFeat(feat, attn_l, attn_r, row_ptr, col_ind, col_ptr, row_ind, permute,
                        negative_slope, attn_drop):
    """out = GATConvFuse(<attn_l,feat>, <attn_r,feat>, graph..., feat, attn_drop) with
    feat [n,h,f], attn_l / attn_r [1,h,f] (gatconv_layer.py:74-80 as one autograd node)."""
    return FusedGATFromFeatFunction.apply(feat, attn_l, attn_r, row_ptr, col_ind, col_ptr, row_ind,
                                          permute, negative_slope, attn_drop)


class FusedGATFromFeatFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, attn_l, attn_r, row_ptr, col_ind, col_ptr, row_ind, permute,
                negative_slope, attn_drop):
        feat, attn_l, attn_r = feat.contiguous(), attn_l.contiguous(), attn_r.contiguous()
        scores = _lo.attn_scores(feat, attn_l, attn_r)
        ctx.fused_scores = scores is not None
        if scores is None:                       # shape outside the kernel's range
            scores = (attn_l * feat).sum(-1), (attn_r * feat).sum(-1)
        attn_row, attn_col = scores
        ctx.f = feat.size(2)
        feat_p = _pad4(feat)                              # rows narrower than four floats run padded (see _pad4)
        out_feat, edge_max, edge_sum, edge_mask, ctx.seed, out_neg, a_neg = _forward_with_dropout(
            attn_row, attn_col, row_ptr, col_ind, negative_slope, feat_p, attn_drop)
        ctx.save_for_backward(row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                              edge_mask, feat, attn_row, attn_col, out_feat, attn_l, attn_r,
                              feat_p if feat_p is not feat else None, out_neg, a_neg)
        ctx.negative_slope = negative_slope
        ctx.attn_drop = attn_drop
        return out_feat if out_feat.size(2) == ctx.f else out_feat[..., :ctx.f]

    @staticmethod
    def backward(ctx, grad_out):
        (row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, feat,
         attn_row, attn_col, out_feat, attn_l, attn_r, feat_p, out_neg, a_neg) = ctx.saved_tensors
        grad_feat, g_row, g_col = _backward_with_dropout(
            ctx, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask,
            feat if feat_p is None else feat_p, attn_row, attn_col,
            _pad_to(grad_out, out_feat.size(2)).contiguous(), out_feat, out_neg, a_neg)
        if grad_feat.size(2) != ctx.f:
            grad_feat = grad_feat[..., :ctx.f].contiguous()
        if ctx.fused_scores:
            g_l, g_r = _lo.attn_scores_backward(feat, attn_l, attn_r, g_row, g_col, grad_feat)
        else:
            g_l = (g_row.unsqueeze(-1) * feat).sum(0, keepdim=True).view_as(attn_l)
            g_r = (g_col.unsqueeze(-1) * feat).sum(0, keepdim=True).view_as(attn_r)
            grad_feat = grad_feat + g_row.unsqueeze(-1) * attn_l + g_col.unsqueeze(-1) * attn_r
        return grad_feat, g_l, g_r, None, None, None, None, None, None, None


def _activation_code(activation):
    """(code, param) of a callable the epilogue kernel implements, else None."""
    import torch.nn.functional as F
    from torch import nn
    if activation is None:
        return _lo.ACT_NONE, 0.0
    if activation in (F.relu, torch.relu) or isinstance(activation, nn.ReLU):
        return _lo.ACT_RELU, 0.0
    if activation is F.elu:
        return _lo.ACT_ELU, 1.0
    if isinstance(activation, nn.ELU):
        return _lo.ACT_ELU, float(activation.alpha)
    if activation is F.leaky_relu:
        return _lo.ACT_LEAKY_RELU, 0.01
    if isinstance(activation, nn.LeakyReLU):
        return _lo.ACT_LEAKY_RELU, float(activation.negative_slope)
    return None


def GATEpilogue(rst, residual, bias, activation):
    """act(rst + residual + bias) in one pass (gatconv_layer.py:82-90); ``activation`` is a
    callable; ones the kernel does not implement are applied by torch after the fused add."""
    code = _activation_code(activation)
    if code is None:
        return activation(GATEpilogueFunction.apply(rst, residual, bias, _lo.ACT_NONE, 0.0))
    if residual is None and bias is None and code[0] == _lo.ACT_NONE:
        return rst
    return GATEpilogueFunction.apply(rst, residual, bias, code[0], code[1])


class GATEpilogueFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rst, residual, bias, act, act_param):
        rst = rst.contiguous()
        if residual is not None:
            residual = residual.contiguous().view_as(rst)
        out = _lo.epilogue(rst, residual, None if bias is None else bias.contiguous(), act, act_param)
        ctx.save_for_backward(out)
        ctx.act, ctx.act_param = act, act_param
        ctx.has_res, ctx.has_bias = residual is not None, bias is not None
        ctx.bias_shape = None if bias is None else bias.shape
        return out

    @staticmethod
    def backward(ctx, grad):
        (out,) = ctx.saved_tensors
        grad_pre, grad_bias = _lo.epilogue_backward(out, grad.contiguous(), ctx.act, ctx.act_param,
                                                    ctx.has_bias and ctx.needs_input_grad[2])
        if grad_bias is not None:
            grad_bias = grad_bias.view(ctx.bias_shape)
        return grad_pre, (grad_pre if ctx.has_res else None), grad_bias, None, None
