"""Semantic head (upsnet/models/fcn.py:29-113): shared deformable-conv subnet on P2..P5, bilinear
upsampling to P2 size, concat, 1x1 score, bilinear x4.

MI355X note: the subnet's weights are shared by the four levels, so each deformable layer runs as ONE
fused launch over all levels (csrc/deform_conv.hip) with bias + ReLU in the epilogue and no column
buffer; only the small 3x3 offset convolutions stay per level.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import ops
from ..config.config import config
from . import hipconv
from ..operators.modules.deform_conv import DeformConv, DeformConvWithOffset
from ..operators.modules.roialign import RoIAlign


class FCNSubNet(nn.Module):

    def __init__(self, in_channels, out_channels, num_layers, deformable_group=1, dilation=1, with_norm='none'):
        super(FCNSubNet, self).__init__()
        assert with_norm == 'none'
        assert num_layers >= 2
        self.num_layers = num_layers
        self.conv = nn.ModuleList()
        for i in range(num_layers):
            conv = []
            if i == num_layers - 2:
                conv.append(DeformConvWithOffset(in_channels, out_channels, kernel_size=3, stride=1, padding=dilation, dilation=dilation))
                in_channels = out_channels
            else:
                conv.append(DeformConvWithOffset(in_channels, in_channels, kernel_size=3, stride=1, padding=dilation, dilation=dilation))
            conv.append(nn.ReLU(inplace=True))
            self.conv.append(nn.Sequential(*conv))
        self._packed = {}
        self.taps = None   # parity tests: set to a dict to record every layer's predicted offsets (tests/test_trunk_gpu.py)
        self.initialize()

    def initialize(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.fill_(0)
                m.bias.data.fill_(0)
            elif isinstance(m, DeformConv):
                nn.init.kaiming_normal_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.fill_(0)

    def forward(self, x):
        for i in range(self.num_layers):
            x = self.conv[i](x)
        return x

    def _wpack(self, i, dc):
        key = (dc.weight.data_ptr(), dc.weight._version, ops.dcn_precision())
        if i not in self._packed or self._packed[i][0] != key:
            self._packed[i] = (key, ops.pack_dcn_weight(dc.weight.detach()))
        return self._packed[i][1]

    def forward_levels(self, feats):
        """All FPN levels through the shared subnet: one fused DCN launch per layer."""
        xs = list(feats)
        for i in range(self.num_layers):
            layer = self.conv[i][0]
            dc = layer.conv
            if not ops.fused_dcn_supported(dc.in_channels, dc.out_channels, dc.deformable_groups, dc.groups, dc.padding, dc.stride, dc.dilation):
                xs = [self.conv[i](x) for x in xs]
                continue
            offsets = hipconv.conv_multi(layer.conv_offset, xs)
            if self.taps is not None:
                self.taps.setdefault('offsets', []).append([o.detach().clone() for o in offsets])
            ys = ops.deform_conv_fused(xs, offsets, self._wpack(i, dc), dc.bias, dc.in_channels, dc.out_channels,
                                       dc.kernel_size, dc.stride, dc.padding, dc.dilation, relu=True)
            hipconv._trace('dcn', module=dc, xs=xs, offsets=offsets, outs=ys, relu=True, form='dcn_fused multi' + (' bf16' if ops.dcn_precision() == 'bf16' else ''))
            xs = ys
        return xs


class FCNHead(nn.Module):

    def __init__(self, in_channels, num_classes, num_layers, with_norm='none', with_roi_loss=False, upsample_rate=4):
        super(FCNHead, self).__init__()
        self.fcn_subnet = FCNSubNet(in_channels, 128, num_layers, with_norm=with_norm)
        self.upsample_rate = upsample_rate
        self.score = nn.Conv2d(512, num_classes, 1)
        if with_roi_loss:
            self.roipool = RoIAlign(config.network.mask_size, config.network.mask_size, 1 / 4.0)
        self.initialize()

    def initialize(self):
        nn.init.normal_(self.score.weight.data, 0, 0.01)
        self.score.bias.data.zero_()

    def _score_parts(self):
        """Per-level column blocks of the 1x1 score weight, packed for the MFMA conv kernel (cached)."""
        w = self.score.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, '_score_pack', None) is None or self._score_pack[0] != key:
            c = w.shape[1] // 4
            self._score_pack = (key, [ops.pack_conv_weight(w.detach()[:, l * c:(l + 1) * c].contiguous()) for l in range(4)])
        return self._score_pack[1]

    def forward_score(self, fpn_p2, fpn_p3, fpn_p4, fpn_p5, commute=True):
        """fcn_score only ([1,S,H/4,W/4]); the x4 upsampling is fused into the panoptic kernel by the caller.
        commute=True evaluates conv1x1(cat(up(y_l))) as sum_l up(W_l y_l) (both linear): four small products at the
        levels' own resolution + one combine kernel instead of three 128-channel upsample passes, a 512-channel concat
        and the wide conv. Same value up to fp32 summation order."""
        ys = self.fcn_subnet.forward_levels([fpn_p2, fpn_p3, fpn_p4, fpn_p5])
        S = self.score.out_channels
        if (commute and ys[0].is_cuda and ys[0].shape[0] == 1 and ys[0].shape[2] % 8 == 0 and ys[0].shape[3] % 8 == 0 and
                all(y.shape[1] * 4 == self.score.in_channels and ops.conv_supported(y.shape[1], 1, 1, 1, (1, 1)) for y in ys)):
            packs = self._score_parts()
            # (one launch for the four levels, each with its own column block of the score weight)
            parts = ops.conv2d_nhwc_multiw(ys, [wp for wp, _ in packs], packs[0][1], S, 1, 1, 0)
            return ops.fcn_score_combine(parts, self.score.bias)
        fpn_p2, fpn_p3, fpn_p4, fpn_p5 = ys
        fpn_p3 = F.interpolate(fpn_p3, None, 2, mode='bilinear', align_corners=False)
        fpn_p4 = F.interpolate(fpn_p4, None, 4, mode='bilinear', align_corners=False)
        fpn_p5 = F.interpolate(fpn_p5, None, 8, mode='bilinear', align_corners=False)
        return hipconv.conv(self.score, torch.cat([fpn_p2, fpn_p3, fpn_p4, fpn_p5], dim=1))

    def forward(self, fpn_p2, fpn_p3, fpn_p4, fpn_p5, roi=None):
        fpn_p2, fpn_p3, fpn_p4, fpn_p5 = self.fcn_subnet.forward_levels([fpn_p2, fpn_p3, fpn_p4, fpn_p5])
        fpn_p3 = F.interpolate(fpn_p3, None, 2, mode='bilinear', align_corners=False)
        fpn_p4 = F.interpolate(fpn_p4, None, 4, mode='bilinear', align_corners=False)
        fpn_p5 = F.interpolate(fpn_p5, None, 8, mode='bilinear', align_corners=False)
        feat = torch.cat([fpn_p2, fpn_p3, fpn_p4, fpn_p5], dim=1)
        score = hipconv.conv(self.score, feat)
        ret = {'fcn_score': score, 'fcn_feat': feat}
        if self.upsample_rate != 1:
            ret['fcn_output'] = F.interpolate(score, None, self.upsample_rate, mode='bilinear', align_corners=False)
        if roi is not None:
            ret['fcn_roi_score'] = self.score(self.roipool(feat, roi))
        return ret
