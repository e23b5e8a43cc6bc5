"""MaskRemoval (upsnet/operators/modules/mask_removal.py:23-93) on the device.

forward(mask_rois [m,4], cls_prob [m], mask_prob [m,1,28,28], cls_idx [m], im_shape) ->
(keep_inds int64 [k], mask_energy [1,k,H,W]). `select` is the variant used by the fused panoptic head:
it returns the selection only (no [k,H,W] planes are materialised).
"""
import torch.nn as nn

from ... import ops
from ...config.config import config


class MaskRemoval(nn.Module):

    def __init__(self, fraction_threshold=0.3):
        super(MaskRemoval, self).__init__()
        self.fraction_threshold = fraction_threshold

    def select(self, mask_rois, cls_prob, mask_prob, cls_idx, im_shape, num_dev=None):
        """Selection only: (keep_inds, num_keep, real_keep) device tensors. num_dev: device count of valid rows when the inputs
        are fixed-capacity buffers."""
        return ops.mask_removal(mask_rois.detach(), cls_prob.detach(), mask_prob.detach(), cls_idx,
                                config.dataset.num_classes - 1, im_shape, self.fraction_threshold, m_dev=num_dev)

    def forward(self, mask_rois, cls_prob, mask_prob, cls_idx, im_shape):
        keep, num, real = self.select(mask_rois, cls_prob, mask_prob, cls_idx, im_shape)
        k = int(num.item())
        energy = ops.mask_paste(mask_rois.detach(), mask_prob.detach(), keep, num, real, k, im_shape)
        return keep[:k], energy
