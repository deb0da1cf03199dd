"""k-nearest-neighbour graph in front of the fused EdgeConv.

The reference takes this step from DGL (``KNNGraph(k)(h)`` then ``src, dst = g.edges()``,
script/train/train_edgeconv.py:17,49-52); DGL is a third-party dependency outside the reference
tree.  ``knn_graph`` returns what the reference passes on as ``src_ind``: for point r of the
flattened batch, ``src_ind[r*k:(r+1)*k]`` = ids of its k nearest points inside its own cloud
(itself first), ascending distance, ties -> lower id.  HIP kernel: csrc/knn.hip.
"""
import torch

from dgnn_amd import _lib as L


def knn_graph(x, k):
    """x: [batch, points, dim] (or [points, dim] = one cloud) float32 GPU tensor -> int32
    ``src_ind`` [batch*points*k]."""
    if isinstance(x, torch.Tensor) and x.dim() == 2:
        x = x.unsqueeze(0)
    L.f32(x, "x", 3)
    batch, points, dim = x.shape
    k = int(k)
    with torch.cuda.device(x.device):
        src_ind = torch.empty(batch * points * k, dtype=torch.int32, device=x.device)
        rc = L.lib().dgnn_knn_graph(batch, points, dim, k, L.p(x), L.p(src_ind), L.stream_of(x))
    L.check(rc, "dgnn_knn_graph")
    return src_ind
