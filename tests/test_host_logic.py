"""Host-side logic that needs no GPU: the split weight gradient of the GAT layer's projection."""
import torch


def test_projection_weight_gradient_split_over_row_blocks():
    """dgnn_amd/layers/_project.py: feat^T @ grad as a batched GEMM over row blocks + remainder
    equals the single GEMM (float64: exact up to summation order), also through autograd."""
    from dgnn_amd.layers._project import _Project, project, weight_grad
    torch.manual_seed(0)
    for n in (1000, (1 << 16) + 13, 3 * 8192 * 3):
        x = torch.randn(n, 10, dtype=torch.float64, requires_grad=True)
        w = torch.randn(10, 6, dtype=torch.float64, requires_grad=True)
        g = torch.randn(n, 6, dtype=torch.float64)
        assert torch.allclose(weight_grad(x.detach(), g), x.detach().t() @ g, rtol=1e-12, atol=1e-9)
        (project if n >= (1 << 16) else _Project.apply)(x, w).backward(g)
        gx, gw = x.grad.clone(), w.grad.clone()
        x.grad = w.grad = None
        (x @ w).backward(g)
        assert torch.allclose(gx, x.grad) and torch.allclose(gw, w.grad, rtol=1e-12, atol=1e-9)
    # 3-d inputs fall back to torch.matmul
    assert project(torch.ones(2, 3, 4), torch.ones(4, 5)).shape == (2, 3, 5)


def test_examples_cross_entropy_equals_the_library_loss():
    """examples/common.cross_entropy: the parallel formulation used above 8192 rows (torch's
    nll_loss reduces in one workgroup) has the value and the gradient of nn.CrossEntropyLoss."""
    import os
    import sys
    import torch
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples"))
    from common import cross_entropy
    gen = torch.Generator().manual_seed(0)
    for rows in (100, 20000):
        x = torch.randn((rows, 47), generator=gen, dtype=torch.float64, requires_grad=True)
        y = torch.randint(0, 47, (rows,), generator=gen)
        a, b = cross_entropy(x, y), torch.nn.CrossEntropyLoss()(x, y)
        ga, = torch.autograd.grad(a, x)
        gb, = torch.autograd.grad(b, x)
        assert torch.allclose(a, b, rtol=1e-12, atol=0) and torch.allclose(ga, gb, rtol=1e-10, atol=1e-15)
