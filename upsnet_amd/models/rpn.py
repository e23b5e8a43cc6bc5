"""RPN head (upsnet/models/rpn.py:26-57): 3x3 conv + ReLU, 1x1 objectness (sigmoid), 1x1 box deltas."""
import torch
import torch.nn as nn

from . import hipconv


class RPN(nn.Module):
    def __init__(self, num_anchors=15, input_dim=256, with_norm='none'):
        super(RPN, self).__init__()
        assert with_norm == 'none'
        self.num_anchors = num_anchors
        self.conv_proposal = nn.Sequential(nn.Conv2d(input_dim, input_dim, 3, padding=1), nn.ReLU(inplace=True))
        self.cls_score = nn.Conv2d(input_dim, self.num_anchors, 1)
        self.bbox_pred = nn.Conv2d(input_dim, self.num_anchors * 4, 1)
        self.initialize()

    def initialize(self):
        for m in [self.conv_proposal[0], self.cls_score, self.bbox_pred]:
            nn.init.normal_(m.weight.data, 0, 0.01)
            if m.bias is not None:
                m.bias.data.zero_()

    def forward(self, data):
        x = hipconv.conv(self.conv_proposal[0], data, relu=True)
        cls_score = hipconv.conv(self.cls_score, x)
        bbox_pred = hipconv.conv(self.bbox_pred, x)
        cls_prob = torch.sigmoid(cls_score)
        return cls_score, bbox_pred, cls_prob


def rpn_forward_levels(rpn, feats):
    """All pyramid levels through the shared RPN head: 3 launches (3x3 conv; both 1x1 heads together; ONE sigmoid over the flat
    buffer that backs the head outputs of every level) instead of 4 per level. The probabilities are returned as channel slices of
    NHWC maps; the proposal kernels consume them in place (strided)."""
    xs = hipconv.conv_multi(rpn.conv_proposal[0], feats, relu=True)
    (scores, boxes), flat, outs = hipconv.conv_multi_cat([rpn.cls_score, rpn.bbox_pred], xs, return_flat=True)
    if flat is None:
        return scores, boxes, [torch.sigmoid(s) for s in scores]
    sig = torch.sigmoid(flat)     # (also over the 12 box-delta channels of each pixel: 2.6 M elements, one launch)
    probs, A = [], rpn.cls_score.out_channels
    for o in outs:
        view = torch.as_strided(sig, o.shape, o.stride(), o.storage_offset() - flat.storage_offset())
        probs.append(view[:, :A])
    return scores, boxes, probs
