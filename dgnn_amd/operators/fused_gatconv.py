"""Autograd operator of the fused GAT path.  API of dgNN/operators/fused_gatconv.py:6-34."""
import torch

from ..native import fused_gatconv as fused_gat


def GATConvFuse(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute, negative_slope,
                in_feat, attn_drop):
    """dgNN/operators/fused_gatconv.py:6-7 (same argument order)."""
    return FusedGATFunction.apply(attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute,
                                  negative_slope, in_feat, attn_drop)


def GATConvFuse_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """dgNN/operators/fused_gatconv.py:9-10."""
    return fused_gat.gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat)


class FusedGATFunction(torch.autograd.Function):
    """dgNN/operators/fused_gatconv.py:12-34.  Differences, both invisible to callers: the
    forward output is saved too (it turns the backward's row term into one dense dot product,
    removing a whole gather pass), and no [nnz,h] mask is kept when attn_drop == 0."""

    @staticmethod
    def forward(ctx, attn_row, attn_col, row_ptr, col_ind, col_ptr, row_ind, permute,
                negative_slope, in_feat, attn_drop):
        attn_row = attn_row.contiguous()
        attn_col = attn_col.contiguous()
        in_feat = in_feat.contiguous()
        out_feat, edge_max, edge_sum, edge_mask = fused_gat.gat_forward(
            attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop)
        ctx.save_for_backward(row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum,
                              edge_mask, in_feat, attn_row, attn_col, out_feat)
        ctx.negative_slope = negative_slope
        ctx.attn_drop = attn_drop
        return out_feat

    @staticmethod
    def backward(ctx, grad_out):
        (row_ptr, col_ind, col_ptr, row_ind, permute, edge_max, edge_sum, edge_mask, in_feat,
         attn_row, attn_col, out_feat) = ctx.saved_tensors
        grad_out = grad_out.contiguous()                       # fused_gatconv.py:26
        grad_feat, grad_attn_row, grad_attn_col = fused_gat.gat_backward(
            ctx.negative_slope, ctx.attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute,
            edge_max, edge_sum, edge_mask, in_feat, attn_row, attn_col, grad_out, out_feat=out_feat)
        # same tuple layout as fused_gatconv.py:34
        return grad_attn_row, grad_attn_col, None, None, None, None, None, None, grad_feat, None
