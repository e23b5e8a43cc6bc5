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
    """All pyramid levels through the shared RPN head: 2 launches (3x3 conv; both 1x1 heads together) instead of 3 per level."""
    xs = hipconv.conv_multi(rpn.conv_proposal[0], feats, relu=True)
    scores, boxes = hipconv.conv_multi_cat([rpn.cls_score, rpn.bbox_pred], xs)
    return scores, boxes, [torch.sigmoid(s) for s in scores]
