"""Drop-in for the reference's pybind module ``mhtranspose``
(dgNN/src/csr2csc/mhtranspose.cc:20-23)."""
import torch

from dgnn_amd import _lib as L


def mhtranspose(permute, attention):
    """-> out[i,:] = attention[permute[i],:]   (mhtranspose.cc:10-18)."""
    L.i32(permute, "permute"), L.f32(attention, "attention", 2)
    nnz, h = permute.size(0), attention.size(1)
    with torch.cuda.device(attention.device):
        out = torch.empty((nnz, h), dtype=torch.float32, device=attention.device)
        rc = L.lib().dgnn_mhtranspose(nnz, h, L.p(permute), L.p(attention), L.p(out),
                                      L.stream_of(attention))
    L.check(rc, "dgnn_mhtranspose")
    return out
