"""DeformConvFunction with the reference's signature (upsnet/operators/functions/deform_conv.py:26-57).

forward(data, offset, weight, bias, in_channels, out_channels, kernel_size, stride, padding, dilation,
groups, deformable_groups). Inference only (the backward kernels are out of scope).
Fast path: the fused NHWC MFMA kernel (no column buffer). Shapes the fused kernel does not cover
(deformable_groups > 1, Cin % 32 != 0, ...) take the reference's own structure -- per-image HIP
im2col into a caller-allocated column buffer + one GEMM -- through the NCHW drop-in entry point.
"""
import numpy as np
import torch
from torch.autograd import Function

from ... import ops


class _DeformConvCuda(object):
    """Stand-in for the reference's pybind module `deform_conv_cuda` (deform_conv_cuda.cpp:107-112)."""
    deform_im2col = staticmethod(ops.deform_im2col)


deform_conv_cuda = _DeformConvCuda()


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


class DeformConvFunction(Function):

    @staticmethod
    def forward(ctx, data, offset, weight, bias, in_channels, out_channels, kernel_size, stride, padding, dilation,
                groups, deformable_groups):
        if not data.is_cuda or not offset.is_cuda or not weight.is_cuda or (bias is not None and not bias.is_cuda):
            raise Exception('not implemented')
        kernel_size, stride, padding, dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        B, C, H, W = data.shape
        Ho, Wo = ops.out_hw(H, W, kernel_size, padding, stride, dilation)
        if ops.fused_dcn_supported(in_channels, out_channels, deformable_groups, groups):
            wpack = ops.pack_dcn_weight(weight)
            outs = [ops.deform_conv_fused([data[i:i + 1]], [offset[i:i + 1]], wpack, bias, in_channels, out_channels,
                                          kernel_size, stride, padding, dilation)[0] for i in range(B)]
            return outs[0] if B == 1 else torch.cat(outs, 0)
        # reference structure: im2col + mm per image (functions/deform_conv.py:44-56)
        data, offset = data.float().contiguous(), offset.float().contiguous()
        kdim = int(in_channels * np.prod(kernel_size))
        col_buffer = data.new_zeros((kdim, Ho, Wo))
        output = data.new_zeros((B, out_channels, Ho, Wo))
        for i in range(B):
            deform_conv_cuda.deform_im2col(data[i], offset[i], tuple(data.shape), tuple(col_buffer.shape), kernel_size,
                                           padding, stride, dilation, 1, deformable_groups, col_buffer)
            output[i] = torch.mm(weight.reshape(-1, kdim), col_buffer.view(kdim, -1)).view(out_channels, Ho, Wo)
        if bias is not None:
            output += bias.view(1, -1, 1, 1)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        raise NotImplementedError("upsnet_amd implements the inference path only")
