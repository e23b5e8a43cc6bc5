"""Deformable-convolution layers, v2 / modulated (API of upsnet/operators/modules/mod_deform_conv.py:24-81).

The reference module cannot be imported (it names a package that does not exist); what its source states is kept:
the predictor emits 3*k*k*dg planes, the first two thirds are the (y, x) offsets and ``mask = 2 * sigmoid(last third)``
(:60-63). State-dict keys: ``conv_offset_mask.*`` and ``conv.{weight,bias}``.
"""
import torch
import torch.nn as nn

from ..functions.mod_deform_conv import ModDeformConvFunction
from .deform_conv import _DeformLayer, _zero_init_predictor


class ModDeformConv(_DeformLayer):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias)

    def forward(self, data, offset, mask):
        return ModDeformConvFunction.apply(data, offset, mask, self.weight, self.bias, *self._geometry())


class ModDeformConvWithOffsetMask(nn.Module):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.conv_offset_mask = _zero_init_predictor(in_channels, 3 * kernel_size * kernel_size * deformable_groups)
        self.conv = ModDeformConv(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                                  groups=groups, deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        planes = self.conv_offset_mask(x)
        n_off = 2 * planes.shape[1] // 3
        return self.conv(x, planes[:, :n_off].contiguous(), 2.0 * torch.sigmoid(planes[:, n_off:]))
