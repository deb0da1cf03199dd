# Same exports as the reference's dgNN/layers/__init__.py (without its import-time print, :1).
from .gatconv_layer import GATConv
from .gmmconv_layer import GMMConv
from .edgeconv_layer import EdgeConv
from .knn_graph import KNNGraph      # replaces dgl.nn.pytorch.KNNGraph (train_edgeconv.py:4)
