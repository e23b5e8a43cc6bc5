"""ModDeformConvFunction (DCN v2) with the reference's signature
(upsnet/operators/functions/mod_deform_conv.py:25-59): forward(data, offset, mask, weight, bias,
in_channels, out_channels, kernel_size, stride, padding, dilation, groups, deformable_groups).
Dead code in the reference (no caller); built for API parity of the north-star operator list.
"""
import numpy as np
import torch
from torch.autograd import Function

from ... import ops
from .deform_conv import _pair


class _ModDeformConvCuda(object):
    mod_deform_im2col = staticmethod(ops.mod_deform_im2col)


mod_deform_conv_cuda = _ModDeformConvCuda()


class ModDeformConvFunction(Function):

    @staticmethod
    def forward(ctx, data, offset, mask, weight, bias, in_channels, out_channels, kernel_size, stride, padding, dilation,
                groups, deformable_groups):
        if not data.is_cuda or not offset.is_cuda or not mask.is_cuda or not weight.is_cuda:
            raise Exception('not implemented')
        kernel_size, stride, padding, dilation = _pair(kernel_size), _pair(stride), _pair(padding), _pair(dilation)
        B, C, H, W = data.shape
        Ho, Wo = ops.out_hw(H, W, kernel_size, padding, stride, dilation)
        if ops.fused_dcn_supported(in_channels, out_channels, deformable_groups, groups):
            wpack = ops.pack_dcn_weight(weight)
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
        raise NotImplementedError("upsnet_amd implements the inference path only")
