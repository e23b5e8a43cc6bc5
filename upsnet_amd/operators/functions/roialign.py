"""RoIAlignFunction with the reference's instantiate-then-call surface
(upsnet/operators/functions/roialign.py:21-43): RoIAlignFunction(ph, pw, spatial_scale, sampling_ratio=2)(features, rois).

The reference is a legacy (non-static) autograd Function, illegal in torch >= 1.5; this is a thin
callable with the same constructor and call signature. NCHW-contiguous features take the NCHW
drop-in kernel and return a contiguous [N,C,PH,PW] tensor exactly like the reference; channels_last
features take the coalesced NHWC kernel and return a channels_last tensor (same values).
"""
import torch

from ... import ops


class _RoIAlignCuda(object):
    """Stand-in for the pybind module `roi_align_cuda` (roi_align_cuda.cpp:114-118)."""

    @staticmethod
    def roi_align_forward(pooled_height, pooled_width, sampling_ratio, spatial_scale, features, rois, output):
        if rois.shape[1] != 5:
            return 0
        output.copy_(ops.roi_align_nchw(features, rois, pooled_height, pooled_width, spatial_scale, sampling_ratio))
        return 1

    roi_align_backward = staticmethod(ops.roi_align_backward)


roi_align_cuda = _RoIAlignCuda()


class _RoIAlignGrad(torch.autograd.Function):
    """What makes the callable below differentiable under today's autograd (the reference relied on the legacy
    instance-Function protocol): forward = the native forward, backward = roi_align_backward into a zeroed map."""

    @staticmethod
    def forward(ctx, features, rois, fn):
        ctx.fn = fn
        return fn._forward(features, rois)

    @staticmethod
    def backward(ctx, grad_output):
        return ctx.fn.backward(grad_output)[0], None, None


class RoIAlignFunction(object):
    def __init__(self, pooled_height, pooled_width, spatial_scale, sampling_ratio=2):
        self.pooled_width = int(pooled_width)
        self.pooled_height = int(pooled_height)
        self.spatial_scale = float(spatial_scale)
        self.sampling_ratio = sampling_ratio
        self.feature_size = None

    def forward(self, features, rois):
        if features.requires_grad and torch.is_grad_enabled():
            return _RoIAlignGrad.apply(features, rois, self)
        return self._forward(features, rois)

    def _forward(self, features, rois):
        if not features.is_cuda:
            raise Exception('not implemented')
        self.feature_size = features.size()
        self.rois = rois
        if features.dim() == 4 and features.shape[1] % 4 == 0 and not features.is_contiguous() and \
                features.is_contiguous(memory_format=torch.channels_last):
            return ops.roi_align_nhwc(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale,
                                      self.sampling_ratio)
        return ops.roi_align_nchw(features, rois, self.pooled_height, self.pooled_width, self.spatial_scale,
                                  self.sampling_ratio)

    __call__ = forward

    def backward(self, grad_output):
        # functions/roialign.py:45-54
        assert self.feature_size is not None and grad_output.is_cuda
        batch_size, num_channels, data_height, data_width = self.feature_size
        grad_input = grad_output.new_zeros((batch_size, num_channels, data_height, data_width), dtype=torch.float32)
        roi_align_cuda.roi_align_backward(self.pooled_height, self.pooled_width, self.sampling_ratio, self.spatial_scale,
                                          grad_output.detach().float().contiguous(), self.rois.detach().float().contiguous(),
                                          grad_input)
        return grad_input, None
