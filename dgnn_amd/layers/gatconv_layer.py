"""GAT layer on the fused operator.  Mirrors dgNN/layers/gatconv_layer.py:7-92: same
constructor, parameter names (state_dict compatible: W, attn_l, attn_r, bias, res_fc.weight),
initialisation and forward signature."""
import torch
from torch import nn

from ..native import gat_layer_ops
from ._project import project
from ..operators.fused_gatconv import GATConvFuseFromFeat, GATConvFuse_inference, GATEpilogue


class GATConv(nn.Module):
    def __init__(self, in_feats, out_feats, num_heads, feat_drop=0., attn_drop=0.,
                 negative_slope=0.2, residual=False, activation=None, bias=True):
        super().__init__()
        self.in_feats = in_feats
        self.out_feats = out_feats
        self.num_heads = num_heads
        self.W = nn.Parameter(torch.empty(in_feats, out_feats * num_heads))
        self.attn_l = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.attn_r = nn.Parameter(torch.empty(1, num_heads, out_feats))
        self.negative_slope = negative_slope
        self.attn_drop = attn_drop
        self.feat_drop = nn.Dropout(feat_drop)
        if bias:
            self.bias = nn.Parameter(torch.empty(num_heads * out_feats))
        else:
            self.register_buffer('bias', None)
        if residual:
            if in_feats != out_feats * num_heads:
                self.res_fc = nn.Linear(in_feats, out_feats * num_heads, bias=False)
            else:
                self.res_fc = nn.Identity()
        else:
            self.register_buffer('res_fc', None)
        self.reset_parameters()
        self.activation = activation

    def reset_parameters(self):
        # gatconv_layer.py:46-66: xavier normal with the relu gain, zero bias
        gain = nn.init.calculate_gain('relu')
        nn.init.xavier_normal_(self.W, gain=gain)
        nn.init.xavier_normal_(self.attn_l, gain=gain)
        nn.init.xavier_normal_(self.attn_r, gain=gain)
        if self.bias is not None:
            nn.init.constant_(self.bias, 0)
        if isinstance(self.res_fc, nn.Linear):
            nn.init.xavier_normal_(self.res_fc.weight, gain=gain)

    def forward(self, row_ptr, col_ind, col_ptr, row_ind, permute, feat):
        h = project(feat, self.W).view(-1, self.num_heads, self.out_feats)
        h = self.feat_drop(h)
        # attn_l is the DESTINATION (row) term, attn_r the SOURCE term (gatconv_layer.py:74-75;
        # the opposite of DGL's naming, test/func_test/func_gat_dgl.py:62)
        # Both score terms come out of one pass over h (csrc/layer.hip) instead of two
        # multiply + sum chains; in training their gradient is folded into the fused operator's
        # grad_feat in place.
        if not self.training:
            h = h.contiguous()
            scores = gat_layer_ops.attn_scores(h.detach(), self.attn_l.detach().contiguous(),
                                               self.attn_r.detach().contiguous())
            if scores is None:
                scores = ((self.attn_l * h).sum(dim=-1).contiguous(),
                          (self.attn_r * h).sum(dim=-1).contiguous())
            rst = GATConvFuse_inference(scores[0], scores[1], row_ptr, col_ind, self.negative_slope, h)
        else:
            rst = GATConvFuseFromFeat(h, self.attn_l, self.attn_r, row_ptr, col_ind, col_ptr, row_ind,
                                      permute, self.negative_slope, self.attn_drop)
        res = None
        if self.res_fc is not None:
            # the reference applies res_fc to h (gatconv_layer.py:83), which only type-checks
            # when in_feats == out_feats*num_heads; apply it to the layer input in the Linear case
            src = h if isinstance(self.res_fc, nn.Identity) else feat
            res = self.res_fc(src).view(-1, self.num_heads, self.out_feats)
        # residual + bias + activation in one pass (gatconv_layer.py:82-90)
        return GATEpilogue(rst, res, self.bias, self.activation if self.activation else None)
