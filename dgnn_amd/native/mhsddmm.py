"""Drop-in for the reference's pybind module ``mhsddmm`` (dgNN/src/sddmm/mhsddmm.cc:30-33)."""
import torch

from dgnn_amd import _lib as L


def mhsddmm(rowptr, colind, grad, feature):
    """-> out [E, H] = <grad[row(e),h,:], feature[col(e),h,:]>   (mhsddmm.cc:11-28)."""
    L.i32(rowptr, "rowptr"), L.i32(colind, "colind")
    L.f32(grad, "grad", 3), L.f32(feature, "feature", 3)
    v, nnz = rowptr.size(0) - 1, colind.size(0)
    h, f = feature.size(1), feature.size(2)
    if grad.size(1) != h or grad.size(2) != f:
        raise L.DgnnError("mhsddmm: grad and feature must both be [*, H, F]")
    with torch.cuda.device(grad.device):
        out = torch.empty((nnz, h), dtype=torch.float32, device=grad.device)
        rc = L.lib().dgnn_mhsddmm(v, f, h, nnz, L.p(rowptr), L.p(colind), L.p(grad), L.p(feature),
                                  L.p(out), L.stream_of(grad))
    L.check(rc, "dgnn_mhsddmm")
    return out
