"""Drop-in for the reference's pybind module ``fused_gmmconv``
(dgNN/src/fused_gmmconv/fused_gmmconv.cpp:59-64)."""
import torch

from dgnn_amd import _lib as L


def gmmconv(csrptr, colind, node_feat, pseudo, mu, inv_sigma):
    """-> out [nodes, K, embed]   (fused_gmmconv.cpp:37-47)."""
    L.i32(csrptr, "csrptr"), L.i32(colind, "colind")
    L.f32(node_feat, "node_feat", 3), L.f32(pseudo, "pseudo", 2), L.f32(mu, "mu", 2)
    L.f32(inv_sigma, "inv_sigma", 2)
    nodes, nnz = csrptr.size(0) - 1, colind.size(0)
    k, dim, embed = mu.size(0), mu.size(1), node_feat.size(2)   # fused_gmmconv_kernel.cu:107-111
    if node_feat.size(1) != k or pseudo.size(0) != nnz or pseudo.size(1) != dim:
        raise L.DgnnError("gmmconv: node_feat [n,K,F], pseudo [nnz,dim], mu [K,dim] disagree")
    with torch.cuda.device(node_feat.device):
        out = torch.empty((nodes, k, embed), dtype=torch.float32, device=node_feat.device)
        nb = L.lib().dgnn_gmm_forward_workspace_bytes(nodes, nnz, k, embed)
        ws = L.workspace(nb, node_feat)
        rc = L.lib().dgnn_gmm_forward(nodes, nnz, k, dim, embed, L.p(csrptr), L.p(colind),
                                      L.p(node_feat), L.p(pseudo), L.p(mu), L.p(inv_sigma), L.p(out),
                                      L.p(ws), nb, L.stream_of(node_feat))
    L.check(rc, "dgnn_gmm_forward")
    return out


def Gaussian(pseudo, mu, inv_sigma):
    """-> gauss [E, K]   (fused_gmmconv.cpp:29-35)."""
    L.f32(pseudo, "pseudo", 2), L.f32(mu, "mu", 2), L.f32(inv_sigma, "inv_sigma", 2)
    e, k, dim = pseudo.size(0), mu.size(0), mu.size(1)
    with torch.cuda.device(pseudo.device):
        out = torch.empty((e, k), dtype=torch.float32, device=pseudo.device)
        rc = L.lib().dgnn_gaussian(e, k, dim, L.p(pseudo), L.p(mu), L.p(inv_sigma), L.p(out),
                                   L.stream_of(pseudo))
    L.check(rc, "dgnn_gaussian")
    return out


def GaussianBp(pseudo, mu, inv_sigma, grad_out):
    """-> [pseudo_out, mu_out, sigma_out]   (fused_gmmconv.cpp:49-57, order of
    fused_gmmconv_kernel.cu:143).  Analytic gradient; see include/dgnn_hip.h."""
    L.f32(pseudo, "pseudo", 2), L.f32(mu, "mu", 2), L.f32(inv_sigma, "inv_sigma", 2)
    L.f32(grad_out, "grad_out", 2)
    e, k, dim = pseudo.size(0), mu.size(0), mu.size(1)
    with torch.cuda.device(pseudo.device):
        po = torch.empty((e, dim), dtype=torch.float32, device=pseudo.device)
        mo = torch.empty((k, dim), dtype=torch.float32, device=pseudo.device)
        so = torch.empty((k, dim), dtype=torch.float32, device=pseudo.device)
        nb = L.lib().dgnn_gaussian_backward_workspace_bytes(e, k, dim)
        ws = L.workspace(nb, pseudo)
        rc = L.lib().dgnn_gaussian_backward(e, k, dim, L.p(pseudo), L.p(mu), L.p(inv_sigma),
                                            L.p(grad_out), L.p(po), L.p(mo), L.p(so), L.p(ws), nb,
                                            L.stream_of(pseudo))
    L.check(rc, "dgnn_gaussian_backward")
    return [po, mo, so]


def gmm_backward_fused(colptr, rowind, permute, node_feat, pseudo, mu, inv_sigma, grad_out):
    """-> (grad_node_feat [nodes,K,embed], grad_gauss [E,K] in CSR order), or None when the shape
    is outside the one-pass kernel's range.  Not a reference export: it stands for the
    mhsddmm + mhtranspose + gmmconv(CSC) sequence of dgNN/operators/fused_gmmconv.py:21-24."""
    L.i32(colptr, "colptr"), L.i32(rowind, "rowind"), L.i32(permute, "permute")
    L.f32(node_feat, "node_feat", 3), L.f32(pseudo, "pseudo", 2), L.f32(grad_out, "grad_out", 3)
    L.f32(mu, "mu", 2), L.f32(inv_sigma, "inv_sigma", 2)
    nodes, nnz = colptr.size(0) - 1, rowind.size(0)
    kernels, embed, dim = node_feat.size(1), node_feat.size(2), pseudo.size(1)
    if grad_out.shape != node_feat.shape or pseudo.size(0) != nnz or node_feat.size(0) != nodes:
        raise L.DgnnError("gmm_backward_fused: grad_out / node_feat / pseudo shapes disagree with the graph")
    nb = L.lib().dgnn_gmm_backward_workspace_bytes(nodes, nnz, kernels, embed)
    if nb == 0:
        return None
    with torch.cuda.device(node_feat.device):
        grad_feat = torch.empty_like(node_feat)
        grad_gauss = torch.empty((nnz, kernels), dtype=torch.float32, device=node_feat.device)
        ws = L.workspace(nb, node_feat)
        rc = L.lib().dgnn_gmm_backward(nodes, nnz, kernels, dim, embed, L.p(colptr), L.p(rowind), L.p(permute),
                                       L.p(node_feat), L.p(pseudo), L.p(mu), L.p(inv_sigma), L.p(grad_out),
                                       L.p(grad_feat), L.p(grad_gauss), L.p(ws), nb, L.stream_of(node_feat))
    if rc == -2:
        return None
    L.check(rc, "dgnn_gmm_backward")
    return grad_feat, grad_gauss
