"""FPNRoIAlign (upsnet/operators/modules/fpn_roi_align.py:20-62).

The reference copies the rois to the host, assigns FPN levels in numpy, launches one ROIAlign per level
(with a dummy ROI for empty levels) and re-orders with an argsort. Here the level is computed on the
device inside ONE launch over all four levels and the output is written directly in ROI order.

``forward(feat_list, rois)`` returns [N,C,PH,PW] -- NCHW-contiguous by default (what the reference
returns, so ``pool_feat.view(N, -1)`` works for a drop-in caller); ``channels_last=True`` keeps the
kernel's native NHWC output (no transpose), which is what the model's heads consume.
"""
from torch.nn.modules.module import Module

from ... import ops


class FPNRoIAlign(Module):
    def __init__(self, pooled_height, pooled_width, spatial_scale, with_expand=False, channels_last=False):
        super(FPNRoIAlign, self).__init__()
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = spatial_scale
        self.with_expand = with_expand
        self.channels_last = channels_last

    def forward(self, feat, rois, num_rois_dev=None):
        out = ops.fpn_roi_align(list(feat[:4]), rois.detach(), self.pooled_height, self.pooled_width,
                                list(self.spatial_scale), 2, num_rois_dev)
        if self.channels_last:
            return out
        return out.contiguous()  # logical NCHW, NCHW-contiguous (layout plumbing only)
