"""ModDeformConv / ModDeformConvWithOffsetMask (upsnet/operators/modules/mod_deform_conv.py:24-81).

The reference module is unreachable (it imports a non-existent package); the semantics restated here
are the ones its source describes: ``offset_mask`` is split in thirds, ``offset = cat(first two)``,
``mask = sigmoid(third) * 2`` (:60-63), then the modulated deformable convolution.
"""
import math

import torch
import torch.nn as nn
from torch.nn.modules.utils import _pair
from torch.nn.parameter import Parameter

from ..functions.mod_deform_conv import ModDeformConvFunction
from .deform_conv import _param_device


class ModDeformConv(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModDeformConv, self).__init__()
        assert in_channels % groups == 0 and out_channels % groups == 0
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        dev = _param_device()
        self.weight = Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size, device=dev))
        if bias:
            self.bias = Parameter(torch.empty(out_channels, device=dev))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        n = self.in_channels
        for k in self.kernel_size:
            n *= k
        stdv = 1. / math.sqrt(n)
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)

    def forward(self, data, offset, mask):
        return ModDeformConvFunction.apply(data, offset, mask, self.weight, self.bias, self.in_channels,
                                           self.out_channels, self.kernel_size, self.stride, self.padding,
                                           self.dilation, self.groups, self.deformable_groups)


class ModDeformConvWithOffsetMask(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super(ModDeformConvWithOffsetMask, self).__init__()
        self.conv_offset_mask = nn.Conv2d(in_channels, kernel_size * kernel_size * 3 * deformable_groups,
                                          kernel_size=3, stride=1, padding=1)
        self.conv_offset_mask.weight.data.zero_()
        self.conv_offset_mask.bias.data.zero_()
        self.conv = ModDeformConv(in_channels, out_channels, kernel_size=kernel_size, stride=stride, padding=padding,
                                  dilation=dilation, groups=groups, deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        offset_mask = self.conv_offset_mask(x)
        o1, o2, m = torch.chunk(offset_mask, 3, dim=1)
        offset = torch.cat((o1, o2), dim=1)
        mask = torch.sigmoid(m) * 2
        return self.conv(x, offset, mask)
