"""Autograd operator of fused GMMConv.  API of dgNN/operators/fused_gmmconv.py:9-27."""
import torch

from ..native import fused_gmmconv as gmm
from ..native import mhsddmm
from ..native import mhtranspose
from ._pad import pad4, pad_to, unpad


def GmmConvFuse(rowptr, colind, colptr, rowind, permute, node_feat, pseudo, mu, sigma):
    """dgNN/operators/fused_gmmconv.py:9-10; `sigma` is the INVERSE sigma, as in the reference."""
    return GmmFuseFunction.apply(rowptr, colind, colptr, rowind, permute, node_feat, pseudo, mu, sigma)


GMMConvFuse = GmmConvFuse   # spelling used by BASELINE.json's north_star


class GmmFuseFunction(torch.autograd.Function):
    """dgNN/operators/fused_gmmconv.py:12-27: forward gmmconv on CSR; backward = SDDMM for the
    Gaussian gradient, pseudo permuted to CSC order, gmmconv on the CSC for the feature
    gradient, Gaussian backward for pseudo / mu / inv_sigma."""

    @staticmethod
    def forward(ctx, rowptr, colind, colptr, rowind, permute, node_feat, pseudo, mu, sigma):
        ctx.f = node_feat.size(2)
        node_feat, pseudo = pad4(node_feat.contiguous()), pseudo.contiguous()   # odd widths run padded
        mu, sigma = mu.contiguous(), sigma.contiguous()
        out = gmm.gmmconv(rowptr, colind, node_feat, pseudo, mu, sigma)
        ctx.save_for_backward(rowptr, colind, colptr, rowind, permute, node_feat, pseudo, mu, sigma)
        return unpad(out, ctx.f)

    @staticmethod
    def backward(ctx, grad_out):
        rowptr, colind, colptr, rowind, permute, node_feat, pseudo, mu, sigma = ctx.saved_tensors
        grad_out = pad_to(grad_out, node_feat.size(2)).contiguous()      # the forward's width
        # one gather pass over the CSC yields both the feature gradient and the Gaussian
        # gradient; shapes outside its range take the reference's three-call sequence
        fused = gmm.gmm_backward_fused(colptr, rowind, permute, node_feat, pseudo, mu, sigma, grad_out)
        if fused is not None:
            node_feat_grad, gaussian_grad = fused
        else:
            gaussian_grad = mhsddmm.mhsddmm(rowptr, colind, grad_out, node_feat)
            pseudo_tran = mhtranspose.mhtranspose(permute, pseudo)
            node_feat_grad = gmm.gmmconv(colptr, rowind, grad_out, pseudo_tran, mu, sigma)
        pseudo_out, mu_out, sigma_out = gmm.GaussianBp(pseudo, mu, sigma, gaussian_grad)
        return None, None, None, None, None, unpad(node_feat_grad, ctx.f), pseudo_out, mu_out, sigma_out
