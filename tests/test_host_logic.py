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
