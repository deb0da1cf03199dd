"""Drop-in for the reference's pybind module ``format_conversion``
(dgNN/src/convert/convert.cpp:37-41)."""
import torch

from dgnn_amd import _lib as L


def csr2csc_int(rowptr, colind, n_cols=None):
    """-> (colptr, rowind, permute), all int32, bit-exact stable transpose on the device."""
    L.i32(rowptr, "rowptr"), L.i32(colind, "colind")
    m, nnz = rowptr.size(0) - 1, colind.size(0)
    n = m if n_cols is None else int(n_cols)                  # reference assumes square, convert.cpp:38
    with torch.cuda.device(rowptr.device):
        colptr = torch.empty(n + 1, dtype=torch.int32, device=rowptr.device)
        rowind = torch.empty(nnz, dtype=torch.int32, device=rowptr.device)
        permute = torch.empty(nnz, dtype=torch.int32, device=rowptr.device)
        nb = L.lib().dgnn_csr2csc_workspace_bytes(m, n, nnz)
        ws = L.workspace(nb, rowptr)
        rc = L.lib().dgnn_csr2csc(m, n, nnz, L.p(rowptr), L.p(colind), L.p(colptr), L.p(rowind),
                                  L.p(permute), L.p(ws), nb, L.stream_of(rowptr))
    L.check(rc, "dgnn_csr2csc")
    return colptr, rowind, permute


def csr2csc(rowptr, colind, csr_data):
    """-> [colptr, rowind, cscval]   (convert.cpp:16-25): csr_data (float32, [nnz]) is carried
    to CSC order, exactly what cusparseCsr2cscEx2 does with the values."""
    L.f32(csr_data, "csr_data", 1)
    colptr, rowind, permute = csr2csc_int(rowptr, colind)
    return [colptr, rowind, csr_data[permute.long()]]


def coo2csr(cooRowInd, m):
    """-> rowptr [m+1]   (convert.cpp:27-35); cooRowInd sorted ascending."""
    L.i32(cooRowInd, "cooRowInd")
    m = int(m)
    with torch.cuda.device(cooRowInd.device):
        rowptr = torch.empty(m + 1, dtype=torch.int32, device=cooRowInd.device)
        rc = L.lib().dgnn_coo2csr(m, cooRowInd.size(0), L.p(cooRowInd), L.p(rowptr),
                                  L.stream_of(cooRowInd))
    L.check(rc, "dgnn_coo2csr")
    return rowptr
