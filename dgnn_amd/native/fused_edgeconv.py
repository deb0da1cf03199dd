"""Drop-in for the reference's pybind module ``fused_edgeconv``
(dgNN/src/fused_edgeconv/fused_edgeconv.cpp:47-51)."""
import torch

from dgnn_amd import _lib as L


def edgeconv_forward(k, src_ind, h_src, h_dst, return_int_idx=False):
    """-> [out_feat, max_idx]; max_idx is a float32 tensor holding ids, as in the reference
    (fused_edgeconv_kernel.cu:46,61).  ``return_int_idx`` additionally returns the exact int32
    copy (ids >= 2^24 are not representable in float32)."""
    L.i32(src_ind, "src_ind"), L.f32(h_src, "h_src", 2), L.f32(h_dst, "h_dst", 2)
    m, f = h_src.size(0), h_src.size(1)                       # fused_edgeconv_kernel.cu:55-56
    k = int(k)
    if h_dst.size(0) != m or h_dst.size(1) != f or src_ind.numel() != m * k:
        raise L.DgnnError("edgeconv_forward: h_src/h_dst must be [m,f] and src_ind [m*k]")
    with torch.cuda.device(h_src.device):
        out = torch.empty((m, f), dtype=torch.float32, device=h_src.device)
        idx_f = torch.empty((m, f), dtype=torch.float32, device=h_src.device)
        idx_i = torch.empty((m, f), dtype=torch.int32, device=h_src.device) if return_int_idx else None
        rc = L.lib().dgnn_edgeconv_forward(m, f, k, L.p(src_ind), L.p(h_src), L.p(h_dst), L.p(out),
                                           L.p(idx_f), L.p(idx_i), L.stream_of(h_src))
    L.check(rc, "dgnn_edgeconv_forward")
    return [out, idx_f, idx_i] if return_int_idx else [out, idx_f]


def edgeconv_backward(grad_out, max_idx):
    """-> grad_src   (fused_edgeconv.cpp:32-45); max_idx int32."""
    L.f32(grad_out, "grad_out", 2), L.i32(max_idx, "max_idx", 2)
    m, f = grad_out.size(0), grad_out.size(1)
    with torch.cuda.device(grad_out.device):
        grad_src = torch.empty((m, f), dtype=torch.float32, device=grad_out.device)
        rc = L.lib().dgnn_edgeconv_backward(m, f, L.p(max_idx), L.p(grad_out), L.p(grad_src),
                                            L.stream_of(grad_out))
    L.check(rc, "dgnn_edgeconv_backward")
    return grad_src
