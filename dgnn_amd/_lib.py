"""ctypes view of ``libdgnn_hip.so`` (C ABI: ``include/dgnn_hip.h``) and the error class of the
package.

The operators reach the library through the C++ extension modules of ``dgnn_amd/native/``
(csrc/torch_binding.cpp); this module serves what needs the raw ABI from Python: the symbol /
signature table the ABI tests check against the header, the workspace queries, and the
kernel-timing hook bench.py reads (``last_kernel_ms``).  There is NO CPU or eager fallback: if
the library is missing, or a tensor is not on a GPU, the call raises ``DgnnError``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DGNN_HIP_LIB selects another build of the same library (kernel tuning A/B runs)
LIB_PATH = os.environ.get("DGNN_HIP_LIB") or os.path.join(_HERE, "csrc", "libdgnn_hip.so")

_vp = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_sz = ctypes.c_size_t
_u64 = ctypes.c_uint64

# name -> (restype, argtypes); must list every symbol include/dgnn_hip.h declares.
SIGNATURES = {
    "dgnn_abi_version": (_i, []),
    "dgnn_error_string": (ctypes.c_char_p, [_i]),
    "dgnn_debug_last_kernel_ms": (_i, [_i, _vp]),
    "dgnn_debug_last_kernel_name": (ctypes.c_char_p, [_i]),
    "dgnn_debug_kernel_launches": (ctypes.c_longlong, [_i]),
    "dgnn_gat_forward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_gat_forward": (_i, [_i, _i, _i, _i, _f, _f] + [_vp] * 11 + [_vp, _sz, _vp]),
    "dgnn_gat_forward_noalloc": (_i, [_i, _i, _i, _i, _f, _f] + [_vp] * 9 + [_vp]),
    "dgnn_gat_inference": (_i, [_i, _i, _i, _i, _f] + [_vp] * 6 + [_vp, _sz, _vp]),
    "dgnn_gat_forward_tiles_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_gat_forward_tiles": (_i, [_i, _i, _i, _i, _f] + [_vp] * 6 + [_i, _i] + [_vp] * 3 + [_vp, _sz, _vp]),
    "dgnn_gat_row_dot": (_i, [_i, _i, _i] + [_vp] * 3 + [_vp]),
    "dgnn_gat_row_stats": (_i, [_i, _i, _i, _f] + [_vp] * 9 + [_vp]),
    "dgnn_gat_backward_workspace_bytes": (_sz, [_i, _i, _i, _i, _i, _i]),
    "dgnn_gat_backward": (_i, [_i, _i, _i, _i, _i, _f, _f] + [_vp] * 19 + [_vp, _sz, _vp]),
    "dgnn_gat_backward_packed": (_i, [_i, _i, _i, _i, _i, _f, _f] + [_vp] * 13 + [_vp, _sz, _vp]),
    "dgnn_dropout_mask": (_i, [_i, _i, _u64, _u64, _vp, _vp]),
    "dgnn_pack_rows": (_i, [_i, _i, _i, _vp, _vp, _vp, _vp]),
    "dgnn_gat_forward_seeded": (_i, [_i, _i, _i, _i, _f, _f] + [_vp] * 5 + [_u64, _u64] + [_vp] * 5 + [_vp, _sz, _vp]),
    "dgnn_gat_backward_seeded": (_i, [_i, _i, _i, _i, _i, _f, _f] + [_vp] * 7 + [_u64, _u64] + [_vp] * 11 + [_vp, _sz, _vp]),
    "dgnn_gat_forward_seeded_dev": (_i, [_i, _i, _i, _i, _f, _f] + [_vp] * 5 + [_vp] + [_vp] * 5 + [_vp, _sz, _vp]),
    "dgnn_gat_backward_seeded_dev": (_i, [_i, _i, _i, _i, _i, _f, _f] + [_vp] * 7 + [_vp] + [_vp] * 11 + [_vp, _sz, _vp]),
    "dgnn_gat_saved_floats": (_sz, [_i, _i, _i]),
    "dgnn_gat_forward_saved": (_i, [_i, _i, _i, _i, _f, _f] + [_vp] * 5 + [_u64, _u64] + [_vp] * 4 + [_vp, _sz, _vp]),
    "dgnn_gat_backward_saved": (_i, [_i, _i, _i, _i, _i, _f, _f] + [_vp] * 15 + [_vp, _sz, _vp]),
    "dgnn_spmm_csr_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_spmm_csr": (_i, [_i, _i, _i, _i] + [_vp] * 5 + [_vp, _sz, _vp]),
    "dgnn_mhsddmm": (_i, [_i, _i, _i, _i] + [_vp] * 5 + [_vp]),
    "dgnn_mhtranspose": (_i, [_i, _i] + [_vp] * 3 + [_vp]),
    "dgnn_gmm_forward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_gmm_forward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 7 + [_vp, _sz, _vp]),
    "dgnn_gmm_backward_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_gmm_backward": (_i, [_i, _i, _i, _i, _i] + [_vp] * 10 + [_vp, _sz, _vp]),
    "dgnn_gmm_forward_sum": (_i, [_i, _i, _i, _i, _i] + [_vp] * 7 + [_vp, _sz, _vp]),
    "dgnn_gmm_backward_sum": (_i, [_i, _i, _i, _i, _i] + [_vp] * 10 + [_vp, _sz, _vp]),
    "dgnn_gaussian": (_i, [_i, _i, _i] + [_vp] * 4 + [_vp]),
    "dgnn_gaussian_backward_workspace_bytes": (_sz, [_i, _i, _i]),
    "dgnn_gaussian_backward": (_i, [_i, _i, _i] + [_vp] * 7 + [_vp, _sz, _vp]),
    "dgnn_edgeconv_forward": (_i, [_i, _i, _i] + [_vp] * 6 + [_vp]),
    "dgnn_edgeconv_backward": (_i, [_i, _i] + [_vp] * 3 + [_vp]),
    "dgnn_edgeconv_backward_det_workspace_bytes": (_sz, [_i, _i]),
    "dgnn_edgeconv_backward_det": (_i, [_i, _i, _i] + [_vp] * 4 + [_vp, _sz, _vp]),
    "dgnn_csr2csc_workspace_bytes": (_sz, [_i, _i, _i]),
    "dgnn_csr2csc": (_i, [_i, _i, _i] + [_vp] * 5 + [_vp, _sz, _vp]),
    "dgnn_coo2csr": (_i, [_i, _i] + [_vp] * 2 + [_vp]),
    "dgnn_knn_graph": (_i, [_i, _i, _i, _i] + [_vp] * 2 + [_vp]),
    "dgnn_knn_graph_workspace_bytes": (_sz, [_i, _i, _i, _i]),
    "dgnn_knn_graph_ws": (_i, [_i, _i, _i, _i] + [_vp] * 2 + [_vp, _sz, _vp]),
    "dgnn_gat_attn_scores": (_i, [_i, _i, _i] + [_vp] * 5 + [_vp]),
    "dgnn_gat_attn_scores_backward_workspace_bytes": (_sz, [_i, _i, _i]),
    "dgnn_gat_attn_scores_backward": (_i, [_i, _i, _i] + [_vp] * 8 + [_vp, _sz, _vp]),
    "dgnn_gat_epilogue": (_i, [_i, _i, _i, _f] + [_vp] * 4 + [_vp]),
    "dgnn_gat_epilogue_backward_workspace_bytes": (_sz, [_i, _i]),
    "dgnn_gat_epilogue_backward": (_i, [_i, _i, _i, _f] + [_vp] * 4 + [_vp, _sz, _vp]),
}

_lib = None


class DgnnError(RuntimeError):
    pass


def lib():
    """Load libdgnn_hip.so (once).  Raises if it has not been built: there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DgnnError(
                f"{LIB_PATH} not found: build it first (python -c 'import __graft_entry__ as g; "
                "g.build()' or make -C dgnn_amd/csrc).  dgnn_amd has no CPU/eager fallback.")
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError here = header/library mismatch
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().dgnn_error_string(int(rc))
        raise DgnnError(f"{what} failed with code {rc}: {msg.decode() if msg else '?'}")


def f32(t, name, ndim=None):
    """Reference bindings assert CUDA + float32 + contiguous (fused_gatconv.cpp:15-29)."""
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DgnnError(f"{name} must be a GPU tensor (dgnn_amd has no CPU path)")
    if t.dtype != torch.float32:
        raise DgnnError(f"{name} must be float32, got {t.dtype}")
    if not t.is_contiguous():
        raise DgnnError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise DgnnError(f"{name} must have {ndim} dims, got shape {tuple(t.shape)}")
    return t


def i32(t, name, ndim=1):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise DgnnError(f"{name} must be a GPU tensor (dgnn_amd has no CPU path)")
    if t.dtype != torch.int32:
        raise DgnnError(f"{name} must be int32, got {t.dtype}")
    if not t.is_contiguous():
        raise DgnnError(f"{name} must be contiguous")
    if ndim is not None and t.dim() != ndim:
        raise DgnnError(f"{name} must have {ndim} dims, got shape {tuple(t.shape)}")
    return t


def p(t):
    """Device pointer (or NULL for None / empty tensors)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def stream_of(t):
    return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def workspace(nbytes, like):
    """Scratch from torch's caching allocator (stream-ordered reuse on the current stream)."""
    if nbytes == 0:
        return None
    return torch.empty(int(nbytes), dtype=torch.uint8, device=like.device)


def last_kernel_ms(which):
    """Duration (ms) of the most recent dominant-kernel launch (0 = forward gather, 1 = backward
    gather); needs DGNN_TIMING=1 in the environment before the library is loaded.  None if off."""
    ms = ctypes.c_float(0.0)
    rc = lib().dgnn_debug_last_kernel_ms(int(which), ctypes.byref(ms))
    return float(ms.value) if rc == 0 else None


def kernel_launches(which):
    """Launches of the forward (0) / backward (1) gather kernel since the library was loaded; None
    when timing is off (DGNN_TIMING=1 before the library is loaded)."""
    v = lib().dgnn_debug_kernel_launches(int(which))
    return int(v) if v >= 0 else None


def last_kernel_name(which):
    """Template instantiation of the most recent dominant-kernel launch as rocprofv3 names it
    (e.g. "k_bwd_col_gather<4, 4, 4, 0, false>"); None when timing is off."""
    v = lib().dgnn_debug_last_kernel_name(int(which))
    return v.decode() if v else None
