"""dgnn_amd -- dgNN's fused GNN operators on AMD Instinct MI355X (gfx950).

Public API mirrors the reference package layout (dgNN/operators, dgNN/layers):

    from dgnn_amd.operators import GATConvFuse, GATConvFuse_inference, GmmConvFuse, EdgeConvFuse
    from dgnn_amd.layers import GATConv, GMMConv, EdgeConv

All compute runs in hand-written HIP kernels behind the C ABI of include/dgnn_hip.h
(libdgnn_hip.so).  ``dgnn_amd/native/`` holds the PyTorch-ROCm C++ extension modules (pybind11,
csrc/torch_binding.cpp) with the reference's top-level module names in front of that ABI.
There is no CPU fallback.
"""
import os
import sys

from . import _lib                      # noqa: F401  (DgnnError, the ctypes view of the C ABI)
from ._lib import DgnnError             # noqa: F401

__version__ = "0.1.0"

_NATIVE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "native")


def enable_dropin():
    """Make the reference's TOP-LEVEL native module names importable (``import fused_gatconv``,
    ``fused_gmmconv``, ``fused_edgeconv``, ``mhsddmm``, ``mhtranspose``, ``format_conversion``;
    reference setup.py:29-34) so unmodified dgNN/operators/*.py run on this library."""
    if _NATIVE_DIR not in sys.path:
        sys.path.insert(0, _NATIVE_DIR)
    return _NATIVE_DIR
