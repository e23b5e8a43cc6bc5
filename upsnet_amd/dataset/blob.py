"""Network input blob on the device -- host mirror of BaseDataset.prep_im_for_blob / im_list_to_blob / get_image_blob
(upsnet/dataset/base_dataset.py:120-173, 898-923).

The reference subtracts the pixel means, resizes with cv2 and pads on the host, then copies a 25 MB fp32 blob to the GPU.
Here the uint8 image (6 MB) goes to the device and ONE kernel (csrc/preprocess.hip) produces the padded fp32 blob, either in
the reference's planar layout or as the 4-channel NHWC image the stem convolution consumes directly.
"""
import numpy as np
import torch

from .. import ops
from ..config.config import config


def blob_geometry(height, width, target_size, max_size, stride=None):
    """(im_scale, (Hr, Wr), (Hp, Wp)): base_dataset.py:155-170 (scale capped by max_size; cv2 dsize = cvRound(size*scale))
    and :909-913 (pad to a multiple of rpn_feat_stride[-2])."""
    if stride is None:
        stride = int(config.network.rpn_feat_stride[-2]) if config.network.has_fpn else 1
    im_scale = float(target_size) / float(min(height, width))
    if np.round(im_scale * max(height, width)) > max_size:
        im_scale = float(max_size) / float(max(height, width))
    hr, wr = int(np.rint(height * im_scale)), int(np.rint(width * im_scale))
    hp, wp = int(np.ceil(hr / float(stride)) * stride), int(np.ceil(wr / float(stride)) * stride)
    return im_scale, (hr, wr), (hp, wp)


def get_image_blob(image, target_size=None, max_size=None, pixel_means=None, nhwc4=True, device=None):
    """image: uint8 [H,W,3] BGR (numpy array or tensor, host or device). Returns a dict shaped like the reference's batch:
    {'data': blob, 'im_info': float32 [[Hr, Wr, im_scale]]} with blob = logical [1,4,Hp,Wp] channels_last (nhwc4, default)
    or [1,3,Hp,Wp] NCHW (the reference's layout)."""
    if target_size is None:
        target_size = config.test.scales[0]
    if max_size is None:
        max_size = config.test.max_size
    if pixel_means is None:
        pixel_means = config.network.pixel_means
    if isinstance(image, np.ndarray):
        image = torch.from_numpy(np.ascontiguousarray(image))
    if device is None:
        device = image.device if image.is_cuda else torch.device('cuda', torch.cuda.current_device())
    image = image.to(device, non_blocking=True)
    H, W = image.shape[:2]
    im_scale, resized, padded = blob_geometry(H, W, target_size, max_size)
    blob = ops.prep_image_u8(image, pixel_means, im_scale, resized, padded, nhwc4=nhwc4)
    im_info = np.array([[resized[0], resized[1], im_scale]], dtype=np.float32)
    return {'data': blob, 'im_info': im_info, 'im_scale': im_scale}
