"""Autograd operator of fused EdgeConv.  API of dgNN/operators/fused_edgeconv.py:8-22."""
import torch

from ..native import fused_edgeconv


def EdgeConvFuse(k, src_ind, h_src, h_dst):
    """dgNN/operators/fused_edgeconv.py:8-9."""
    return FusedEdgeConvFunction.apply(k, src_ind, h_src, h_dst)


class FusedEdgeConvFunction(torch.autograd.Function):
    """dgNN/operators/fused_edgeconv.py:11-22.  The arg-max is kept as exact int32 (the
    reference keeps a float tensor and casts it back, :20, which is exact only below 2^24)."""

    @staticmethod
    def forward(ctx, k, src_ind, h_src, h_dst):
        out_feat, _max_idx_f, max_idx = fused_edgeconv.edgeconv_forward(
            k, src_ind, h_src.contiguous(), h_dst.contiguous(), return_int_idx=True)
        ctx.save_for_backward(max_idx, src_ind)
        return out_feat

    @staticmethod
    def backward(ctx, grad_out):
        max_idx, src_ind = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        # gathered over the transposed k-NN lists: fixed summation order, no float atomics
        grad_src = fused_edgeconv.edgeconv_backward(grad_out, max_idx, src_ind=src_ind)
        return None, None, grad_src, grad_out                  # grad_h_dst is grad_out itself, :22
