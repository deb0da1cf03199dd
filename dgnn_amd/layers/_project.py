"""Dense projection ``feat @ W`` of the GAT layer (gatconv_layer.py:71) with a weight gradient
that is not one GEMM reducing over all N rows.

``dW = feat^T @ grad`` is an [in, N] x [N, H*F] product with N = the number of nodes: rocBLAS /
hipBLASLt run it without splitting the reduction, 4.3 ms at N = 2.4 M, in = 100, H*F = 256 on
MI355X -- more than the forward GEMM (1.5 ms) although it moves the same bytes.  Splitting the
rows into blocks, multiplying the blocks as one batched GEMM and summing the partial products
(split-K by hand, fixed order: deterministic) takes 1.2 ms (tools/gemm_probe.py).  Plain torch
ops: the GEMMs stay on the BLAS library, as in the reference.
"""
import torch

_SPLIT_MIN_ROWS = 1 << 16
_ROWS_PER_BLOCK = 8192


def weight_grad(feat, grad):
    """feat [N, in], grad [N, out] -> feat^T @ grad [in, out]."""
    n = feat.size(0)
    if n < _SPLIT_MIN_ROWS:
        return feat.t() @ grad
    b = min(1024, n // _ROWS_PER_BLOCK)
    rows = n // b
    m = rows * b
    part = torch.bmm(feat[:m].view(b, rows, feat.size(1)).transpose(1, 2), grad[:m].view(b, rows, grad.size(1)))
    out = part.sum(0)
    if m != n:
        out.addmm_(feat[m:].t(), grad[m:])
    return out


class _Project(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, weight):
        ctx.save_for_backward(feat, weight)
        return feat @ weight

    @staticmethod
    def backward(ctx, grad):
        feat, weight = ctx.saved_tensors
        grad = grad.contiguous()
        g_feat = grad @ weight.t() if ctx.needs_input_grad[0] else None
        g_weight = weight_grad(feat.contiguous(), grad) if ctx.needs_input_grad[1] else None
        return g_feat, g_weight


def project(feat, weight):
    """feat [N, in] @ weight [in, out] (2-d inputs; anything else falls back to torch.matmul)."""
    if feat.dim() != 2 or weight.dim() != 2 or feat.size(0) < _SPLIT_MIN_ROWS:
        return torch.matmul(feat, weight)          # small inputs: nothing to split, plain autograd
    return _Project.apply(feat, weight)
