"""FPN (upsnet/models/fpn.py:25-104): 1x1 laterals, nearest x2 top-down adds, 3x3 outputs,
P6 = stride-2 subsample of P5, optional GAP branch (COCO configs). with_norm='none' only (every
shipped config)."""
import torch.nn as nn
import torch.nn.functional as F

from ..config.config import config
from . import hipconv


class FPN(nn.Module):

    def __init__(self, feature_dim, with_extra_level=True, with_bottom_up_path_aggregation=False, with_norm='none',
                 upsample_method='nearest'):
        super(FPN, self).__init__()
        assert with_norm == 'none', "only fpn_with_norm='none' is used by the reference configs"
        assert upsample_method in ['nearest', 'bilinear']
        self.feature_dim = feature_dim
        self.upsample_method = upsample_method
        if with_extra_level:
            self.fpn_p6 = nn.MaxPool2d(kernel_size=1, stride=2)
        if config.network.fpn_with_gap:
            self.fpn_gap = nn.Linear(2048, feature_dim)
        self.fpn_p5_1x1 = nn.Conv2d(2048, feature_dim, 1)
        self.fpn_p4_1x1 = nn.Conv2d(1024, feature_dim, 1)
        self.fpn_p3_1x1 = nn.Conv2d(512, feature_dim, 1)
        self.fpn_p2_1x1 = nn.Conv2d(256, feature_dim, 1)
        self.fpn_p5 = nn.Conv2d(feature_dim, feature_dim, 3, padding=1)
        self.fpn_p4 = nn.Conv2d(feature_dim, feature_dim, 3, padding=1)
        self.fpn_p3 = nn.Conv2d(feature_dim, feature_dim, 3, padding=1)
        self.fpn_p2 = nn.Conv2d(feature_dim, feature_dim, 3, padding=1)
        self.initialize()

    def initialize(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_uniform_(m.weight.data, a=1)
                if m.bias is not None:
                    m.bias.data.zero_()

    def fpn_upsample(self, x):
        return F.interpolate(x, scale_factor=2, mode=self.upsample_method,
                             align_corners=False if self.upsample_method == 'bilinear' else None)

    def forward(self, res2, res3, res4, res5):
        p5_1x1 = hipconv.conv(self.fpn_p5_1x1, res5)
        if hasattr(self, 'fpn_gap'):
            gap = self.fpn_gap(F.adaptive_avg_pool2d(res5.float(), (1, 1)).flatten(1)).view(-1, self.feature_dim, 1, 1)
            p5_1x1 = p5_1x1 + gap
        # top-down pathway: the lateral 1x1 and the "+ upsampled" add are one kernel (residual epilogue)
        # (and for nearest upsampling the x2 upsample is folded into the residual read: nothing is materialised)
        if self.upsample_method == 'nearest' and all(t.shape[2] % 2 == 0 and t.shape[3] % 2 == 0 for t in (res2, res3, res4)):
            # (bf16 mode: the two large top-down maps are only read by bf16 kernels -- the next lateral's shortcut and the 3x3 output
            # convolution, which rounds its input to bf16 anyway -- and are stored as bf16)
            ad = hipconv.act_dtype()
            p4_plus = hipconv.conv(self.fpn_p4_1x1, res4, residual=p5_1x1, residual_up=True)
            p3_plus = hipconv.conv(self.fpn_p3_1x1, res3, residual=p4_plus, residual_up=True, out_dtype=ad)
            p2_plus = hipconv.conv(self.fpn_p2_1x1, res2, residual=p3_plus, residual_up=True, out_dtype=ad)
        else:
            p4_plus = hipconv.conv(self.fpn_p4_1x1, res4, residual=self.fpn_upsample(p5_1x1))
            p3_plus = hipconv.conv(self.fpn_p3_1x1, res3, residual=self.fpn_upsample(p4_plus))
            p2_plus = hipconv.conv(self.fpn_p2_1x1, res2, residual=self.fpn_upsample(p3_plus))
        p5 = hipconv.conv(self.fpn_p5, p5_1x1)
        p4 = hipconv.conv(self.fpn_p4, p4_plus)
        p3 = hipconv.conv(self.fpn_p3, p3_plus)
        p2 = hipconv.conv(self.fpn_p2, p2_plus)
        if hasattr(self, 'fpn_p6'):
            return p2, p3, p4, p5, self.fpn_p6(p5)
        return p2, p3, p4, p5
