"""Drop-in for the reference's top-level pybind module ``fused_gatconv``
(dgNN/src/fused_gatconv/fused_gatconv.cpp:168-174; built by setup.py:29).

Same function names, argument order and return lists; the work is done by the HIP kernels
behind ``libdgnn_hip.so``.  Put this directory on ``sys.path`` (``dgnn_amd.enable_dropin()``)
and the reference's own ``dgNN/operators/fused_gatconv.py`` imports it unchanged.
"""
import torch

from dgnn_amd import _lib as L


def _dims(row_ptr, col_ind, attn_row, in_feat):
    # shapes are inferred exactly as gat_forward_cuda does (fused_gatconv_kernel.cu:367-370)
    return row_ptr.size(0) - 1, col_ind.size(0), attn_row.size(1), in_feat.size(2)


def _check_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat):
    L.f32(attn_row, "attn_row", 2), L.f32(attn_col, "attn_col", 2), L.f32(in_feat, "in_feat", 3)
    L.i32(row_ptr, "row_ptr"), L.i32(col_ind, "col_ind")
    if in_feat.size(1) != attn_row.size(1) or attn_col.size(1) != attn_row.size(1):
        raise L.DgnnError("attn_row [m,h], attn_col [n,h] and in_feat [n,h,f] disagree on h")
    if attn_col.size(0) != in_feat.size(0):
        raise L.DgnnError("attn_col and in_feat disagree on the number of source nodes")
    if attn_row.size(0) != row_ptr.size(0) - 1:
        raise L.DgnnError("attn_row rows != len(row_ptr) - 1")


def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop,
                edge_mask=None):
    """-> [out_feat, edge_max, edge_sum, edge_mask]   (fused_gatconv.cpp:11-32).

    The reference draws ``edge_mask`` ~ U(0,1] with cuRAND on every call
    (fused_gatconv_kernel.cu:333-341).  Here the mask is drawn with torch's generator only when
    ``attn_drop > 0``; with ``attn_drop == 0`` every edge is kept, no mask is materialised and
    an empty tensor is returned in its place.  ``edge_mask`` may be passed explicitly (tests)."""
    _check_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat)
    m, nnz, h, f = _dims(row_ptr, col_ind, attn_row, in_feat)
    attn_drop = float(attn_drop)
    with torch.cuda.device(in_feat.device):
        out = torch.empty((m, h, f), dtype=torch.float32, device=in_feat.device)
        edge_max = torch.empty((m, h), dtype=torch.float32, device=in_feat.device)
        edge_sum = torch.empty((m, h), dtype=torch.float32, device=in_feat.device)
        if attn_drop > 0.0:
            if edge_mask is None:
                edge_mask = torch.rand((nnz, h), dtype=torch.float32, device=in_feat.device)
            L.f32(edge_mask, "edge_mask", 2)
        else:
            edge_mask = torch.empty((0, h), dtype=torch.float32, device=in_feat.device)
        nb = L.lib().dgnn_gat_forward_workspace_bytes(m, nnz, h, f)
        ws = L.workspace(nb, in_feat)
        rc = L.lib().dgnn_gat_forward(m, nnz, h, f, attn_drop, float(negative_slope), L.p(attn_row),
                                      L.p(attn_col), L.p(row_ptr), L.p(col_ind), L.p(in_feat),
                                      L.p(edge_mask), L.p(edge_max), L.p(edge_sum), L.p(out),
                                      L.p(ws), nb, L.stream_of(in_feat))
    L.check(rc, "dgnn_gat_forward")
    return [out, edge_max, edge_sum, edge_mask]


def gat_forward_tb(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, tile_scheduler):
    """-> [out_feat, edge_max, edge_sum]   (fused_gatconv.cpp:68-95)."""
    _check_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat)
    L.i32(tile_scheduler, "tile_scheduler", 2)
    m, nnz, h, f = _dims(row_ptr, col_ind, attn_row, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty((m, h, f), dtype=torch.float32, device=in_feat.device)
        edge_max = torch.empty((m, h), dtype=torch.float32, device=in_feat.device)
        edge_sum = torch.empty((m, h), dtype=torch.float32, device=in_feat.device)
        nb = L.lib().dgnn_gat_forward_workspace_bytes(m, nnz, h, f)
        ws = L.workspace(nb, in_feat)
        rc = L.lib().dgnn_gat_forward_tiles(m, nnz, h, f, float(negative_slope), L.p(attn_row),
                                            L.p(attn_col), L.p(row_ptr), L.p(col_ind), L.p(in_feat),
                                            L.p(tile_scheduler), tile_scheduler.size(0),
                                            L.p(edge_max), L.p(edge_sum), L.p(out), L.p(ws), nb,
                                            L.stream_of(in_feat))
    L.check(rc, "dgnn_gat_forward_tiles")
    return [out, edge_max, edge_sum]


def make_tile_scheduler(row_ptr, tile=32):
    """The (row, tile_in_row) list ``gat_forward_tb`` takes -- one pair per ``tile`` nonzeros of a
    row -- built on row_ptr's device with index arithmetic instead of the reference's host loop
    over all rows (script/test/test_gat_tb.py:16-26).  -> int32 [n_tiles, 2]."""
    rp = row_ptr.to(torch.int64)
    tiles = (rp[1:] - rp[:-1] + (tile - 1)) // tile
    rid = torch.repeat_interleave(torch.arange(tiles.numel(), device=rp.device), tiles)
    first = torch.cumsum(tiles, 0) - tiles
    tin = torch.arange(rid.numel(), device=rp.device) - first[rid]
    return torch.stack((rid, tin), 1).to(torch.int32).contiguous()


def gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """-> out_feat   (fused_gatconv.cpp:38-60)."""
    _check_fwd(attn_row, attn_col, row_ptr, col_ind, in_feat)
    m, nnz, h, f = _dims(row_ptr, col_ind, attn_row, in_feat)
    with torch.cuda.device(in_feat.device):
        out = torch.empty((m, h, f), dtype=torch.float32, device=in_feat.device)
        nb = L.lib().dgnn_gat_forward_workspace_bytes(m, nnz, h, f)
        ws = L.workspace(nb, in_feat)
        rc = L.lib().dgnn_gat_inference(m, nnz, h, f, float(negative_slope), L.p(attn_row),
                                        L.p(attn_col), L.p(row_ptr), L.p(col_ind), L.p(in_feat),
                                        L.p(out), L.p(ws), nb, L.stream_of(in_feat))
    L.check(rc, "dgnn_gat_inference")
    return out


def gat_row_dot(grad, out_feat):
    """-> S [m,h] = <grad[r,h,:], out_feat[r,h,:]> (the row term of the backward)."""
    L.f32(grad, "grad", 3), L.f32(out_feat, "out_feat", 3)
    m, h, f = grad.size(0), grad.size(1), grad.size(2)
    with torch.cuda.device(grad.device):
        S = torch.empty((m, h), dtype=torch.float32, device=grad.device)
        rc = L.lib().dgnn_gat_row_dot(m, h, f, L.p(grad), L.p(out_feat), L.p(S), L.stream_of(grad))
    L.check(rc, "dgnn_gat_row_dot")
    return S


def gat_backward(negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute, edge_max,
                 edge_sum, edge_mask, in_feat, attn_row, attn_col, grad, out_feat=None, row_dot=None):
    """-> [grad_feat, grad_attn_row, grad_attn_col]   (fused_gatconv.cpp:104-166).

    ``out_feat`` / ``row_dot`` (keywords, not in the reference signature): the saved forward
    output, or the precomputed row term; passing either removes one whole gather pass from the
    backward (see include/dgnn_hip.h).  Rows (m = len(row_ptr)-1) and columns
    (n = len(col_ptr)-1 = in_feat.size(0)) may differ (row/column partitions)."""
    for t, nm in ((row_ptr, "row_ptr"), (col_ind, "col_ind"), (col_ptr, "col_ptr"),
                  (row_ind, "row_ind"), (permute, "permute")):
        L.i32(t, nm)
    L.f32(edge_max, "edge_max", 2), L.f32(edge_sum, "edge_sum", 2), L.f32(in_feat, "in_feat", 3)
    L.f32(attn_row, "attn_row", 2), L.f32(attn_col, "attn_col", 2), L.f32(grad, "grad", 3)
    m, nnz = row_ptr.size(0) - 1, col_ind.size(0)
    n = col_ptr.size(0) - 1
    h, f = in_feat.size(1), in_feat.size(2)          # fused_gatconv_kernel.cu:1049-1050
    if in_feat.size(0) != n or attn_col.size(0) != n:
        raise L.DgnnError("gat_backward: in_feat/attn_col rows must equal len(col_ptr)-1")
    if grad.size(0) != m or attn_row.size(0) != m or edge_max.size(0) != m or edge_sum.size(0) != m:
        raise L.DgnnError("gat_backward: grad/attn_row/edge_max/edge_sum rows must equal len(row_ptr)-1")
    attn_drop = float(attn_drop)
    if attn_drop > 0.0:
        L.f32(edge_mask, "edge_mask", 2)
    if out_feat is not None:
        L.f32(out_feat, "out_feat", 3)
    if row_dot is not None:
        L.f32(row_dot, "row_dot", 2)
    have = 1 if (out_feat is not None or row_dot is not None) else 0
    with torch.cuda.device(in_feat.device):
        grad_feat = torch.empty((n, h, f), dtype=torch.float32, device=in_feat.device)
        grad_attn_row = torch.empty((m, h), dtype=torch.float32, device=in_feat.device)
        grad_attn_col = torch.empty((n, h), dtype=torch.float32, device=in_feat.device)
        nb = L.lib().dgnn_gat_backward_workspace_bytes(m, n, nnz, h, f, have)
        ws = L.workspace(nb, in_feat)
        rc = L.lib().dgnn_gat_backward(
            m, n, nnz, h, f, float(negative_slope), attn_drop, L.p(row_ptr), L.p(col_ind),
            L.p(col_ptr), L.p(row_ind), L.p(permute), L.p(edge_max), L.p(edge_sum),
            L.p(edge_mask) if attn_drop > 0.0 else None, L.p(in_feat), L.p(attn_row), L.p(attn_col),
            L.p(grad), L.p(out_feat), L.p(row_dot), L.p(grad_feat), L.p(grad_attn_row),
            L.p(grad_attn_col), L.p(ws), nb, L.stream_of(in_feat))
    L.check(rc, "dgnn_gat_backward")
    return [grad_feat, grad_attn_row, grad_attn_col]
