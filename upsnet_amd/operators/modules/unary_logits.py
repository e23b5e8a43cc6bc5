"""SegTerm (upsnet/operators/modules/unary_logits.py:69-105) on the device.

forward(cls_indices [k], seg_score [1,S,H,W], boxes [k,5]) -> (seg_energy [1,S_stuff,H,W] (a view),
seg_inst_energy [1,k,H,W]). MaskTerm is training-only in the reference and is not built.
"""
import torch
import torch.nn as nn

from ... import ops
from ...config.config import config


def default_class_mapping(num_seg_classes, num_classes=None):
    num_classes = config.dataset.num_classes if num_classes is None else num_classes
    return dict(zip(range(1, num_classes), range(num_seg_classes - num_classes + 1, num_seg_classes)))


class SegTerm(nn.Module):

    def __init__(self, num_seg_classes, box_scale=1 / 4.0, class_mapping=None, thresh=0.3):
        super(SegTerm, self).__init__()
        self.class_mapping = default_class_mapping(num_seg_classes) if class_mapping is None else class_mapping
        self.num_seg_classes = num_seg_classes
        self.num_inst_classes = len(self.class_mapping)
        self.box_scale = box_scale
        table = torch.zeros((max(self.class_mapping.keys()) + 1,), dtype=torch.int64)
        for c, ch in self.class_mapping.items():
            table[c] = ch
        self.register_buffer('class_map', table, persistent=False)

    def forward(self, cls_indices, seg_score, boxes):
        assert seg_score.shape[0] == 1, "only support batch size = 1"
        seg_energy = seg_score[[0], :-self.num_inst_classes, :, :]
        if cls_indices.numel() == 0:
            return seg_energy, torch.ones_like(seg_energy[[0], [0], :, :]).view(1, 1, seg_energy.shape[2], seg_energy.shape[3]) * -10
        b = boxes[:, 1:].float() * self.box_scale
        seg_inst = ops.seg_term(seg_score, b, cls_indices, self.class_map.to(seg_score.device))
        return seg_energy, seg_inst
