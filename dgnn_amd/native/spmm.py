"""GE-SpMM entry point (new API surface: the reference has no standalone SpMM module,
SURVEY.md section 8a row S1)."""
import torch

from dgnn_amd import _lib as L


def spmm_csr(rowptr, colind, edge_val, in_feat):
    """-> out[r,h,:] = sum_j edge_val[j,h] * in_feat[colind[j],h,:]."""
    L.i32(rowptr, "rowptr"), L.i32(colind, "colind")
    L.f32(edge_val, "edge_val", 2), L.f32(in_feat, "in_feat", 3)
    m, nnz = rowptr.size(0) - 1, colind.size(0)
    h, f = in_feat.size(1), in_feat.size(2)
    if edge_val.size(0) != nnz or edge_val.size(1) != h:
        raise L.DgnnError("spmm_csr: edge_val must be [nnz, H]")
    with torch.cuda.device(in_feat.device):
        out = torch.empty((m, h, f), dtype=torch.float32, device=in_feat.device)
        nb = L.lib().dgnn_spmm_csr_workspace_bytes(m, nnz, h, f)
        ws = L.workspace(nb, in_feat)
        rc = L.lib().dgnn_spmm_csr(m, nnz, h, f, L.p(rowptr), L.p(colind), L.p(edge_val),
                                   L.p(in_feat), L.p(out), L.p(ws), nb, L.stream_of(in_feat))
    L.check(rc, "dgnn_spmm_csr")
    return out
