"""Deformable-convolution layers, v1 (API of upsnet/operators/modules/deform_conv.py:25-78).

What the contract fixes (and nothing else is taken from the reference): the class names, the constructor
argument list, the state-dict keys (``weight`` / ``bias`` on the layer; ``conv_offset.*`` / ``conv.*`` on the
self-contained variant), a U(-s, s) initialisation with s = (Cin*kh*kw)^-1/2 and an all-zero offset predictor.
``_DeformLayer`` carries the shared parameter handling for the v1 layer here and the modulated v2 layer in
``mod_deform_conv.py``; packed-weight caching for the fused HIP kernel lives with the owner of the weight.
"""
import torch
import torch.nn as nn

from ..functions.deform_conv import DeformConvFunction


def _two(v):
    return (int(v[0]), int(v[1])) if isinstance(v, (tuple, list)) else (int(v), int(v))


def _param_device():
    """Parameters are born on the current GPU like the reference's (modules/deform_conv.py:43-46); CPU when there is none."""
    return torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')


class _DeformLayer(nn.Module):
    """Geometry + parameters of a (modulated) deformable convolution; subclasses supply forward()."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias):
        super().__init__()
        for name, total, parts in (('in_channels', in_channels, groups), ('out_channels', out_channels, groups),
                                   ('out_channels', out_channels, deformable_groups)):
            if total % parts:
                raise AssertionError('%s=%d does not split into %d groups' % (name, total, parts))
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _two(kernel_size), _two(stride)
        self.padding, self.dilation = _two(padding), _two(dilation)
        self.groups, self.deformable_groups = groups, deformable_groups
        kh, kw = self.kernel_size
        dev = _param_device()
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, kh, kw, device=dev))
        self.bias = nn.Parameter(torch.empty(out_channels, device=dev)) if bias else None
        self.reset_parameters()

    def reset_parameters(self):
        fan_in = self.in_channels * self.kernel_size[0] * self.kernel_size[1]
        bound = fan_in ** -0.5
        with torch.no_grad():
            for p in (self.weight, self.bias):
                if p is not None:
                    p.uniform_(-bound, bound)

    def _geometry(self):
        return (self.in_channels, self.out_channels, self.kernel_size, self.stride, self.padding, self.dilation,
                self.groups, self.deformable_groups)

    def extra_repr(self):
        return '{}, {}, kernel_size={}, stride={}, padding={}, dilation={}, groups={}, deformable_groups={}'.format(*self._geometry())


class DeformConv(_DeformLayer):

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__(in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups, bias)

    def forward(self, data, offset):
        return DeformConvFunction.apply(data, offset, self.weight, self.bias, *self._geometry())


def _zero_init_predictor(in_channels, planes):
    """3x3 / pad 1 convolution that predicts `planes` sampling channels and starts at exactly zero (= a plain convolution)."""
    conv = nn.Conv2d(in_channels, planes, kernel_size=3, stride=1, padding=1)
    nn.init.zeros_(conv.weight)
    nn.init.zeros_(conv.bias)
    return conv


class DeformConvWithOffset(nn.Module):
    """conv_offset (2*k*k*dg planes from the input itself) feeding a DeformConv."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deformable_groups=1, bias=True):
        super().__init__()
        self.conv_offset = _zero_init_predictor(in_channels, 2 * kernel_size * kernel_size * deformable_groups)
        self.conv = DeformConv(in_channels, out_channels, kernel_size, stride=stride, padding=padding, dilation=dilation,
                               groups=groups, deformable_groups=deformable_groups, bias=bias)

    def forward(self, x):
        return self.conv(x, self.conv_offset(x))
