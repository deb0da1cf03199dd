"""Stand-in for ``dgl.nn.pytorch.KNNGraph`` as the reference's DGCNN uses it
(script/train/train_edgeconv.py:17,49-52): the model only ever takes ``src`` out of the graph,
so calling the module returns that ``src_ind`` tensor directly (int32, on the input's device)."""
from torch import nn

from ..native.knn import knn_graph


class KNNGraph(nn.Module):
    def __init__(self, k):
        super().__init__()
        self.k = k

    def forward(self, x):
        return knn_graph(x.detach().contiguous(), self.k)
