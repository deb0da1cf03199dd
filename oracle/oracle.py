"""ctypes front-end of ``libdgnn_oracle.so`` (numpy in, numpy out).

TEST INFRASTRUCTURE ONLY -- see ``dgnn_oracle.c``.  Every function takes the
same arguments, in the same order, as the reference binding it restates
(cited per function) so parity tests read like the reference's own calls.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libdgnn_oracle.so")

__all__ = [
    "build", "set_threads", "max_threads",
    "gat_forward", "gat_forward_f64", "gat_inference", "gat_backward", "gat_backward_f64",
    "spmm_csr", "spmm_csr_f64", "mhsddmm", "mhsddmm_f64", "mhtranspose",
    "gmmconv", "gmmconv_f64", "gaussian", "gaussian_bp", "gaussian_bp_f64",
    "edgeconv_forward", "edgeconv_backward", "edgeconv_backward_f64",
    "csr2csc", "coo2csr", "knn_graph",
    "gat_attn_scores_f64", "gat_attn_scores_backward_f64", "gat_epilogue_f64", "gat_epilogue_backward_f64",
]


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    src = os.path.join(_HERE, "dgnn_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libdgnn_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def _L():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        _lib.oracle_set_threads(ctypes.c_int(1))   # deterministic by default; bench raises it
    return _lib


def _f32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(np.asarray(a), dtype=np.int32)


def _p(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


_F = ctypes.c_float
_D = ctypes.c_double
_I = ctypes.c_int


def set_threads(n):
    _L().oracle_set_threads(_I(int(n)))


def max_threads():
    return int(_L().oracle_get_max_threads())


# --------------------------------------------------------------------------- GAT
def gat_forward(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop,
                edge_mask=None):
    """fused_gatconv.gat_forward (fused_gatconv.cpp:11-32).  ``edge_mask`` is explicit
    (the reference draws it with cuRAND); None keeps every edge.
    Returns (out_feat, edge_max, edge_sum)."""
    attn_row, attn_col, in_feat = _f32(attn_row), _f32(attn_col), _f32(in_feat)
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], attn_row.shape[1], in_feat.shape[2]
    mask = None if edge_mask is None else _f32(edge_mask)
    out = np.empty((m, h, f), np.float32)
    emax = np.empty((m, h), np.float32)
    esum = np.empty((m, h), np.float32)
    _L().oracle_gat_forward_f32(_I(m), _I(nnz), _I(h), _I(f), _F(attn_drop), _F(negative_slope),
                                _p(attn_row), _p(attn_col), _p(row_ptr), _p(col_ind), _p(in_feat),
                                _p(mask), _p(emax), _p(esum), _p(out))
    return out, emax, esum


def gat_forward_f64(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat, attn_drop,
                    edge_mask=None):
    attn_row, attn_col, in_feat = _f32(attn_row), _f32(attn_col), _f32(in_feat)
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], attn_row.shape[1], in_feat.shape[2]
    mask = None if edge_mask is None else _f32(edge_mask)
    out = np.empty((m, h, f), np.float64)
    emax = np.empty((m, h), np.float64)
    esum = np.empty((m, h), np.float64)
    _L().oracle_gat_forward_f64(_I(m), _I(nnz), _I(h), _I(f), _D(attn_drop), _D(negative_slope),
                                _p(attn_row), _p(attn_col), _p(row_ptr), _p(col_ind), _p(in_feat),
                                _p(mask), _p(emax), _p(esum), _p(out))
    return out, emax, esum


def gat_inference(attn_row, attn_col, row_ptr, col_ind, negative_slope, in_feat):
    """fused_gatconv.gat_inference (fused_gatconv.cpp:38-60)."""
    attn_row, attn_col, in_feat = _f32(attn_row), _f32(attn_col), _f32(in_feat)
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], attn_row.shape[1], in_feat.shape[2]
    out = np.empty((m, h, f), np.float32)
    _L().oracle_gat_inference_f32(_I(m), _I(nnz), _I(h), _I(f), _F(negative_slope), _p(attn_row),
                                  _p(attn_col), _p(row_ptr), _p(col_ind), _p(in_feat), _p(out))
    return out


def gat_backward(negative_slope, attn_drop, row_ptr, col_ind, col_ptr, row_ind, permute,
                 edge_max, edge_sum, edge_mask, in_feat, attn_row, attn_col, grad):
    """fused_gatconv.gat_backward (fused_gatconv.cpp:104-166), same argument order.
    Returns (grad_feat, grad_attn_row, grad_attn_col)."""
    row_ptr, col_ind, col_ptr, row_ind, permute = map(_i32, (row_ptr, col_ind, col_ptr, row_ind, permute))
    edge_max, edge_sum, in_feat, attn_row, attn_col, grad = map(
        _f32, (edge_max, edge_sum, in_feat, attn_row, attn_col, grad))
    mask = None if edge_mask is None else _f32(edge_mask)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], in_feat.shape[1], in_feat.shape[2]
    gf = np.empty((m, h, f), np.float32)
    gr = np.empty((m, h), np.float32)
    gc = np.empty((m, h), np.float32)
    _L().oracle_gat_backward_f32(_I(m), _I(nnz), _I(h), _I(f), _F(negative_slope), _F(attn_drop),
                                 _p(row_ptr), _p(col_ind), _p(col_ptr), _p(row_ind), _p(permute),
                                 _p(edge_max), _p(edge_sum), _p(mask), _p(in_feat), _p(attn_row),
                                 _p(attn_col), _p(grad), _p(gf), _p(gr), _p(gc))
    return gf, gr, gc


def gat_backward_f64(negative_slope, attn_drop, row_ptr, col_ind, edge_mask, in_feat, attn_row,
                     attn_col, grad):
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    in_feat, attn_row, attn_col, grad = map(_f32, (in_feat, attn_row, attn_col, grad))
    mask = None if edge_mask is None else _f32(edge_mask)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], in_feat.shape[1], in_feat.shape[2]
    gf = np.empty((m, h, f), np.float64)
    gr = np.empty((m, h), np.float64)
    gc = np.empty((m, h), np.float64)
    _L().oracle_gat_backward_f64(_I(m), _I(nnz), _I(h), _I(f), _D(negative_slope), _D(attn_drop),
                                 _p(row_ptr), _p(col_ind), _p(mask), _p(in_feat), _p(attn_row),
                                 _p(attn_col), _p(grad), _p(gf), _p(gr), _p(gc))
    return gf, gr, gc


# --------------------------------------------------------------------------- SpMM / SDDMM
def _spmm(fn, dt, row_ptr, col_ind, edge_val, in_feat):
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    edge_val, in_feat = _f32(edge_val), _f32(in_feat)
    m, nnz, h, f = row_ptr.shape[0] - 1, col_ind.shape[0], in_feat.shape[1], in_feat.shape[2]
    out = np.empty((m, h, f), dt)
    fn(_I(m), _I(nnz), _I(h), _I(f), _p(row_ptr), _p(col_ind), _p(edge_val), _p(in_feat), _p(out))
    return out


def spmm_csr(row_ptr, col_ind, edge_val, in_feat):
    """out[r,h,:] = sum_j edge_val[j,h] * in_feat[col_ind[j],h,:]  (SURVEY 8a row S1)."""
    return _spmm(_L().oracle_spmm_csr_f32, np.float32, row_ptr, col_ind, edge_val, in_feat)


def spmm_csr_f64(row_ptr, col_ind, edge_val, in_feat):
    return _spmm(_L().oracle_spmm_csr_f64, np.float64, row_ptr, col_ind, edge_val, in_feat)


def _sddmm(fn, dt, rowptr, colind, grad, feature):
    rowptr, colind = _i32(rowptr), _i32(colind)
    grad, feature = _f32(grad), _f32(feature)
    v, nnz, h, f = rowptr.shape[0] - 1, colind.shape[0], feature.shape[1], feature.shape[2]
    out = np.empty((nnz, h), dt)
    fn(_I(v), _I(f), _I(h), _I(nnz), _p(rowptr), _p(colind), _p(grad), _p(feature), _p(out))
    return out


def mhsddmm(rowptr, colind, grad, feature):
    """mhsddmm.mhsddmm (sddmm/mhsddmm.cc:11-28)."""
    return _sddmm(_L().oracle_mhsddmm_f32, np.float32, rowptr, colind, grad, feature)


def mhsddmm_f64(rowptr, colind, grad, feature):
    return _sddmm(_L().oracle_mhsddmm_f64, np.float64, rowptr, colind, grad, feature)


def mhtranspose(permute, attention):
    """mhtranspose.mhtranspose (csr2csc/mhtranspose.cc:10-18)."""
    permute, attention = _i32(permute), _f32(attention)
    out = np.empty_like(attention)
    _L().oracle_mhtranspose(_I(permute.shape[0]), _I(attention.shape[1]), _p(permute),
                            _p(attention), _p(out))
    return out


# --------------------------------------------------------------------------- GMM
def _gmm(fn, dt, csrptr, colind, node_feat, pseudo, mu, inv_sigma):
    csrptr, colind = _i32(csrptr), _i32(colind)
    node_feat, pseudo, mu, inv_sigma = map(_f32, (node_feat, pseudo, mu, inv_sigma))
    nodes, k, dim, embed = csrptr.shape[0] - 1, mu.shape[0], mu.shape[1], node_feat.shape[2]
    out = np.empty((nodes, k, embed), dt)
    fn(_I(nodes), _I(k), _I(dim), _I(embed), _p(csrptr), _p(colind), _p(node_feat), _p(pseudo),
       _p(mu), _p(inv_sigma), _p(out))
    return out


def gmmconv(csrptr, colind, node_feat, pseudo, mu, inv_sigma):
    """fused_gmmconv.gmmconv (fused_gmmconv.cpp:37-47)."""
    return _gmm(_L().oracle_gmm_forward_f32, np.float32, csrptr, colind, node_feat, pseudo, mu, inv_sigma)


def gmmconv_f64(csrptr, colind, node_feat, pseudo, mu, inv_sigma):
    return _gmm(_L().oracle_gmm_forward_f64, np.float64, csrptr, colind, node_feat, pseudo, mu, inv_sigma)


def gaussian(pseudo, mu, inv_sigma):
    """fused_gmmconv.Gaussian (fused_gmmconv.cpp:29-35)."""
    pseudo, mu, inv_sigma = map(_f32, (pseudo, mu, inv_sigma))
    e, k, dim = pseudo.shape[0], mu.shape[0], mu.shape[1]
    out = np.empty((e, k), np.float32)
    _L().oracle_gaussian_f32(_I(e), _I(k), _I(dim), _p(pseudo), _p(mu), _p(inv_sigma), _p(out))
    return out


def _gbp(fn, dt, pseudo, mu, inv_sigma, grad_out):
    pseudo, mu, inv_sigma, grad_out = map(_f32, (pseudo, mu, inv_sigma, grad_out))
    e, k, dim = pseudo.shape[0], mu.shape[0], mu.shape[1]
    po = np.empty((e, dim), dt)
    mo = np.empty((k, dim), dt)
    so = np.empty((k, dim), dt)
    fn(_I(e), _I(k), _I(dim), _p(pseudo), _p(mu), _p(inv_sigma), _p(grad_out), _p(po), _p(mo), _p(so))
    return po, mo, so


def gaussian_bp(pseudo, mu, inv_sigma, grad_out):
    """fused_gmmconv.GaussianBp (fused_gmmconv.cpp:49-57): the intended gradient, see C file."""
    return _gbp(_L().oracle_gaussian_bp_f32, np.float32, pseudo, mu, inv_sigma, grad_out)


def gaussian_bp_f64(pseudo, mu, inv_sigma, grad_out):
    return _gbp(_L().oracle_gaussian_bp_f64, np.float64, pseudo, mu, inv_sigma, grad_out)


# --------------------------------------------------------------------------- EdgeConv
def edgeconv_forward(k, src_ind, h_src, h_dst):
    """fused_edgeconv.edgeconv_forward (fused_edgeconv.cpp:11-27): (out_feat, max_idx as f32)."""
    src_ind, h_src, h_dst = _i32(src_ind), _f32(h_src), _f32(h_dst)
    m, f = h_src.shape
    out = np.empty((m, f), np.float32)
    idx = np.empty((m, f), np.float32)
    _L().oracle_edgeconv_forward(_I(m), _I(f), _I(int(k)), _p(src_ind), _p(h_src), _p(h_dst),
                                 _p(out), _p(idx))
    return out, idx


def edgeconv_backward(grad_out, max_idx):
    """fused_edgeconv.edgeconv_backward (fused_edgeconv.cpp:32-45); max_idx int32."""
    grad_out, max_idx = _f32(grad_out), _i32(max_idx)
    m, f = grad_out.shape
    out = np.empty((m, f), np.float32)
    _L().oracle_edgeconv_backward_f32(_I(m), _I(f), _p(max_idx), _p(grad_out), _p(out))
    return out


def edgeconv_backward_f64(grad_out, max_idx):
    grad_out, max_idx = _f32(grad_out), _i32(max_idx)
    m, f = grad_out.shape
    out = np.empty((m, f), np.float64)
    _L().oracle_edgeconv_backward_f64(_I(m), _I(f), _p(max_idx), _p(grad_out), _p(out))
    return out


# --------------------------------------------------------------------------- index work
def csr2csc(row_ptr, col_ind, n_cols=None):
    """format_conversion.csr2csc with csr_data = arange (convert.cpp:16-25;
    train_gatconv.py:87-94).  Returns (col_ptr, row_ind, permute), all int32."""
    row_ptr, col_ind = _i32(row_ptr), _i32(col_ind)
    m, nnz = row_ptr.shape[0] - 1, col_ind.shape[0]
    n = m if n_cols is None else int(n_cols)
    col_ptr = np.empty(n + 1, np.int32)
    row_ind = np.empty(nnz, np.int32)
    permute = np.empty(nnz, np.int32)
    _L().oracle_csr2csc(_I(m), _I(n), _I(nnz), _p(row_ptr), _p(col_ind), _p(col_ptr), _p(row_ind),
                        _p(permute))
    return col_ptr, row_ind, permute


def coo2csr(coo_row, m):
    """format_conversion.coo2csr (convert.cpp:27-35): sorted COO rows -> row_ptr."""
    coo_row = _i32(coo_row)
    row_ptr = np.empty(int(m) + 1, np.int32)
    _L().oracle_coo2csr(_I(int(m)), _I(coo_row.shape[0]), _p(coo_row), _p(row_ptr))
    return row_ptr


def knn_graph(x, k):
    """x [batch, points, dim] float32 -> src_ind [batch*points*k] int32 (DGL KNNGraph + g.edges()
    as used at script/train/train_edgeconv.py:49-52; pinned distance chain and tie rule)."""
    x = _f32(x)
    assert x.ndim == 3
    b, pts, dim = x.shape
    out = np.empty(b * pts * int(k), np.int32)
    _L().oracle_knn(_I(b), _I(pts), _I(dim), _I(int(k)), _p(x), _p(out))
    return out


# ---------------------------------------------------------------------------------------------
# GAT layer steps around the fused operator, float64 numpy (dgNN/layers/gatconv_layer.py).
def gat_attn_scores_f64(feat, attn_l, attn_r):
    """gatconv_layer.py:74-75: attn_row = (attn_l * h).sum(-1), attn_col = (attn_r * h).sum(-1)."""
    x = np.asarray(feat, np.float64)
    al = np.asarray(attn_l, np.float64).reshape(1, x.shape[1], x.shape[2])
    ar = np.asarray(attn_r, np.float64).reshape(1, x.shape[1], x.shape[2])
    return (al * x).sum(-1), (ar * x).sum(-1)


def gat_attn_scores_backward_f64(feat, attn_l, attn_r, g_row, g_col):
    """Gradients of the two sums above: (d feat, d attn_l [h,f], d attn_r [h,f])."""
    x = np.asarray(feat, np.float64)
    al = np.asarray(attn_l, np.float64).reshape(1, x.shape[1], x.shape[2])
    ar = np.asarray(attn_r, np.float64).reshape(1, x.shape[1], x.shape[2])
    gr = np.asarray(g_row, np.float64)[:, :, None]
    gc = np.asarray(g_col, np.float64)[:, :, None]
    return gr * al + gc * ar, (gr * x).sum(0), (gc * x).sum(0)


def _act(v, activation, param):
    if activation == 1:
        return np.where(v > 0, v, 0.0)
    if activation == 2:
        return np.where(v > 0, v, param * np.expm1(np.minimum(v, 0.0)))
    if activation == 3:
        return np.where(v > 0, v, param * v)
    return v


def gat_epilogue_f64(rst, residual, bias, activation, param):
    """gatconv_layer.py:82-90: rst (+ residual) (+ bias) then activation (0 none, 1 ReLU,
    2 ELU(alpha), 3 LeakyReLU(slope))."""
    v = np.asarray(rst, np.float64).copy()
    if residual is not None:
        v = v + np.asarray(residual, np.float64).reshape(v.shape)
    if bias is not None:
        v = v + np.asarray(bias, np.float64).reshape((1,) + v.shape[1:])
    return _act(v, activation, param)


def gat_epilogue_backward_f64(rst, residual, bias, activation, param, grad):
    """-> (grad_pre, grad_bias): derivative at the pre-activation value."""
    v = gat_epilogue_f64(rst, residual, bias, 0, 0.0)
    if activation == 1:
        d = (v > 0).astype(np.float64)
    elif activation == 2:
        d = np.where(v > 0, 1.0, param * np.exp(np.minimum(v, 0.0)))
    elif activation == 3:
        d = np.where(v > 0, 1.0, param)
    else:
        d = np.ones_like(v)
    gp = np.asarray(grad, np.float64) * d
    return gp, gp.reshape(gp.shape[0], -1).sum(0)
