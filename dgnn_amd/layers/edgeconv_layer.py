"""EdgeConv (DGCNN) layer on the fused operator.  Mirrors dgNN/layers/edgeconv_layer.py:4-41."""
from torch import nn

from ..operators.fused_edgeconv import EdgeConvFuse


class EdgeConv(nn.Module):
    def __init__(self, in_feat, out_feat, batch_norm=False, allow_zero_in_degree=False):
        super().__init__()
        self.batch_norm = batch_norm
        self._allow_zero_in_degree = allow_zero_in_degree
        self.theta = nn.Linear(in_feat, out_feat, bias=False)
        self.phi = nn.Linear(in_feat, out_feat, bias=False)
        if batch_norm:
            # created but never applied, exactly as the reference (edgeconv_layer.py:17-18,34-41)
            self.bn = nn.BatchNorm1d(out_feat)

    def set_allow_zero_in_degree(self, set_value):
        self._allow_zero_in_degree = set_value

    def forward(self, k, src_ind, feat):
        h_theta = self.theta(feat)
        h_phi = self.phi(feat)
        h_src = h_theta
        h_dst = h_phi - h_theta
        return EdgeConvFuse(k, src_ind, h_src, h_dst)
