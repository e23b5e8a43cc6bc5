"""ModDeformConvFunction (DCN v2) with the reference's signature
(upsnet/operators/functions/mod_deform_conv.py:25-59): forward(data, offset, mask, weight, bias,
in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups).
Dead code in the reference (no caller); built, forward and backward, for API parity of the north-star operator list.
"""
import numpy as np
import torch
from torch.autograd import Function

from ... import ops
from .deform_conv import _pair


class _ModDeformConvCuda(object):
    mod_deform_im2col = staticmethod(ops.mod_deform_im2col)
    mod_deform_col2im = staticmethod(ops.mod_deform_col2im)
    mod_deform_col2im_coord = staticmethod(ops.mod_deform_col2im_coord)


mod_deform_conv_cuda = _ModDeformConvCuda()


class ModDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, data, offset, mask, weight, bias, in_channels, out_channels, kernel_size, stride, padding, dilation,
                groups, deformable_groups):
        if not data.is_cuda or not offset.is_cuda or not mask.is_cuda or not weight.is_cuda:
            raise Exception('not implemented')
        kernel_size, stride, padding, dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        if weight.requires_grad or data.requires_grad or offset.requires_grad or mask.requires_grad:
            ctx.save_for_backward(data, offset, mask, weight, bias)
        ctx.conf = (in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups)
        B, C, H, W = data.shape
        Ho, Wo = ops.out_hw(H, W, kernel_size, padding, stride, dilation)
        if ops.fused_dcn_supported(in_channels, out_channels, deformable_groups, groups, padding, stride, dilation):
            wpack = ops.cached_dcn_pack(weight)   # packed once per weight (version-checked), not per call
            outs = [ops.deform_conv_fused([data[i:i + 1]], [offset[i:i + 1]], wpack, bias, in_channels, out_channels,
                                          kernel_size, stride, padding, dilation, masks=[mask[i:i + 1]])[0] for i in range(B)]
            return outs[0] if B == 1 else torch.cat(outs, 0)
        data, offset, mask = data.float().contiguous(), offset.float().contiguous(), mask.float().contiguous()
        kdim = int(in_channels * np.prod(kernel_size))
        col_buffer = data.new_zeros((kdim, Ho, Wo))
        output = data.new_zeros((B, out_channels, Ho, Wo))
        for i in range(B):
            mod_deform_conv_cuda.mod_deform_im2col(data[i], offset[i], mask[i], tuple(data.shape), tuple(col_buffer.shape),
                                                   kernel_size, padding, stride, dilation, deformable_groups, col_buffer)
            output[i] = torch.mm(weight.reshape(-1, kdim), col_buffer.view(kdim, -1)).view(out_channels, Ho, Wo)
        if bias is not None:
            output += bias.view(1, -1, 1, 1)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        # structure of the reference's backward (functions/mod_deform_conv.py:63-100)
        data, offset, mask, weight, bias = ctx.saved_tensors
        if not grad_output.is_cuda:
            raise Exception('not implemented')
        in_channels, out_channels, kernel_size, stride, padding, dilation, deformable_groups = ctx.conf
        data, offset, mask = (t.detach().float().contiguous() for t in (data, offset, mask))
        grad_output = grad_output.detach().float().contiguous()
        w2d = weight.detach().float().reshape(out_channels, -1)
        grad_data, grad_offset, grad_mask = torch.zeros_like(data), torch.zeros_like(offset), torch.zeros_like(mask)
        grad_weight = torch.zeros_like(w2d)
        Ho, Wo = grad_output.shape[2:]
        shape = tuple(data.shape)
        for i in range(shape[0]):
            go = grad_output[i].view(out_channels, -1)
            col_buffer = torch.mm(w2d.t(), go).view(-1, Ho, Wo)
            cshape = tuple(col_buffer.shape)
            mod_deform_conv_cuda.mod_deform_col2im_coord(col_buffer, data[i], offset[i], mask[i], shape, cshape, kernel_size,
                                                         padding, stride, dilation, deformable_groups, grad_offset[i],
                                                         grad_mask[i])
            mod_deform_conv_cuda.mod_deform_col2im(col_buffer, offset[i], mask[i], shape, cshape, kernel_size, padding, stride,
                                                   dilation, deformable_groups, grad_data[i])
            mod_deform_conv_cuda.mod_deform_im2col(data[i], offset[i], mask[i], shape, cshape, kernel_size, padding, stride,
                                                   dilation, deformable_groups, col_buffer)
            grad_weight += torch.mm(go, col_buffer.view(cshape[0], -1).t())
        grad_bias = grad_output.sum(dim=(0, 2, 3)) if bias is not None else None
        return (grad_data, grad_offset, grad_mask, grad_weight.view_as(weight), grad_bias) + (None,) * 8
