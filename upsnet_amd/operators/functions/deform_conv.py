"""DeformConvFunction with the reference's signature (upsnet/operators/functions/deform_conv.py:26-57).

forward(data, offset, weight, bias, in_channels, out_channels, kernel_size, stride, padding, dilation,
groups, deformable_groups). backward follows the reference's structure (:59-95): per image, weight^T x grad_output
into a column buffer, then the col2im_coord / col2im / im2col natives and one GEMM for the weight gradient.
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
    deform_col2im = staticmethod(ops.deform_col2im)
    deform_col2im_coord = staticmethod(ops.deform_col2im_coord)


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
        if weight.requires_grad or data.requires_grad or offset.requires_grad:
            ctx.save_for_backward(data, offset, weight, bias)
        ctx.conf = (in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        B, C, H, W = data.shape
        Ho, Wo = ops.out_hw(H, W, kernel_size, padding, stride, dilation)
        if ops.fused_dcn_supported(in_channels, out_channels, deformable_groups, groups, padding, stride, dilation):
            wpack = ops.cached_dcn_pack(weight)   # packed once per weight (version-checked), not per call
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
        data, offset, weight, bias = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise Exception('not implemented')
        in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups = ctx.conf
        data, offset = data.detach().float().contiguous(), offset.detach().float().contiguous()
        grad_output = grad_output.detach().float().contiguous()
        w2d = weight.detach().float().reshape(out_channels, -1)
        grad_data, grad_offset = torch.zeros_like(data), torch.zeros_like(offset)
        grad_weight = torch.zeros_like(w2d)
        Ho, Wo = grad_output.shape[2:]
        shape = tuple(data.shape)
        for i in range(shape[0]):
            go = grad_output[i].view(out_channels, -1)
            col_buffer = torch.mm(w2d.t(), go).view(-1, Ho, Wo)
            cshape = tuple(col_buffer.shape)
            deform_conv_cuda.deform_col2im_coord(col_buffer, data[i], offset[i], shape, cshape, kernel_size, padding, stride,
                                                 dilation, 1, deformable_groups, grad_offset[i])
            deform_conv_cuda.deform_col2im(col_buffer, offset[i], shape, cshape, kernel_size, padding, stride, dilation, 1,
                                           deformable_groups, grad_data[i])
            deform_conv_cuda.deform_im2col(data[i], offset[i], shape, cshape, kernel_size, padding, stride, dilation, 1,
                                           deformable_groups, col_buffer)
            grad_weight += torch.mm(go, col_buffer.view(cshape[0], -1).t())
        grad_bias = grad_output.sum(dim=(0, 2, 3)) if bias is not None else None
        return (grad_data, grad_offset, grad_weight.view_as(weight), grad_bias) + (None,) * 8
