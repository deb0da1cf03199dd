"""GE-SpMM style CSR aggregation as an autograd operator (new API surface; BASELINE.json's
north_star lists it, the reference only has it inside its fused kernels)."""
import torch

from ..native import mhsddmm, spmm
from ._pad import pad4, pad_to, unpad


def SpMMCsr(rowptr, colind, colptr, rowind, permute, edge_val, in_feat):
    """out[r,h,:] = sum_j edge_val[j,h] * in_feat[colind[j],h,:]; differentiable in edge_val
    and in_feat.  (colptr, rowind, permute) describe the CSC of the same matrix."""
    return SpMMCsrFunction.apply(rowptr, colind, colptr, rowind, permute, edge_val, in_feat)


GESpMM = SpMMCsr


class SpMMCsrFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rowptr, colind, colptr, rowind, permute, edge_val, in_feat):
        ctx.f = in_feat.size(2)
        edge_val, in_feat = edge_val.contiguous(), pad4(in_feat.contiguous())   # rows narrower than four floats run padded
        ctx.save_for_backward(rowptr, colind, colptr, rowind, permute, edge_val, in_feat)
        return unpad(spmm.spmm_csr(rowptr, colind, edge_val, in_feat), ctx.f)

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, colind, colptr, rowind, permute, edge_val, in_feat = ctx.saved_tensors
        grad_out = pad_to(grad_out, in_feat.size(2)).contiguous()        # the forward's width
        grad_val = grad_feat = None
        if ctx.needs_input_grad[5]:
            grad_val = mhsddmm.mhsddmm(rowptr, colind, grad_out, in_feat)
        if ctx.needs_input_grad[6]:
            val_csc = edge_val[permute.long()].contiguous()
            grad_feat = unpad(spmm.spmm_csr(colptr, rowind, val_csc, grad_out), ctx.f)
        return None, None, None, None, None, grad_val, grad_feat
