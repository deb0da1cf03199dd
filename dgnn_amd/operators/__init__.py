# Same exports as the reference's dgNN/operators/__init__.py:2-4, plus the GE-SpMM entry point
# and the `GMMConvFuse` spelling BASELINE.json uses.
from .fused_gatconv import GATConvFuse, GATConvFuse_inference, FusedGATFunction
from .fused_gatconv import GATConvFuseFromFeat, GATEpilogue      # layer-level fusion (not in the reference)
from .fused_gmmconv import GmmConvFuse, GMMConvFuse, GmmFuseFunction
from .fused_gmmconv import GmmConvFuseSum                       # the layer's .sum(1) folded in (not in the reference)
from .fused_edgeconv import EdgeConvFuse, FusedEdgeConvFunction
from .spmm import SpMMCsr, GESpMM, SpMMCsrFunction
