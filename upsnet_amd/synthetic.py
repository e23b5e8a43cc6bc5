"""Synthetic inputs and seeded weights for the benchmark / tests (no datasets or checkpoints offline).

Image: integer-valued U{0..255} BGR minus config.network.pixel_means, fp32 NCHW,
zero-padded to a multiple of 32 (base_dataset.py:910-920), im_info = [[H, W, 1.0]] with the unpadded size.

Weights: the modules' own initialisers under torch.manual_seed(235) (the reference's seed,
upsnet_end2end_train.py:67-69) plus two documented, seeded adjustments that keep the instance branch
from degenerating under random weights: (1) deformable offset convolutions get N(0, offset_std) weights
(the reference zero-initialises them, which would make every DCN a plain conv); (2) rcnn.cls_score is
rescaled so that a realistic number of detections survive the 0.6 panoptic threshold. Every benchmark
number is reported together with n_rois / n_det / n_inst.
"""
import math

import numpy as np
import torch

from .config.config import config


def make_image(height, width, seed=0, device='cpu'):
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (1, 3, height, width), generator=g).float()
    img -= torch.tensor(np.asarray(config.network.pixel_means, dtype=np.float32)).view(1, 3, 1, 1)
    ph, pw = int(math.ceil(height / 32.0) * 32), int(math.ceil(width / 32.0) * 32)
    blob = torch.zeros((1, 3, ph, pw), dtype=torch.float32)
    blob[:, :, :height, :width] = img
    im_info = np.array([[height, width, 1.0]], dtype=np.float32)
    return {'data': blob.to(device), 'im_info': im_info}


def make_image_u8(height, width, seed=0, device='cpu'):
    """The uint8 [H,W,3] source image of make_image(height, width, seed) (same pixel values, HWC)."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randint(0, 256, (1, 3, height, width), generator=g)
    return img[0].permute(1, 2, 0).contiguous().to(torch.uint8).to(device)


def build_model(symbol=None, seed=235, device='cuda', offset_std=0.01, cls_gain=None, pipeline='fused',
                channels_last=True, fold_bn=True):
    """Construct the configured model with seeded synthetic weights, ready for inference."""
    from .models.resnet_upsnet import resnet_50_upsnet, resnet_101_upsnet
    ctor = {'resnet_50_upsnet': resnet_50_upsnet, 'resnet_101_upsnet': resnet_101_upsnet}[symbol or config.symbol]
    torch.manual_seed(seed)
    model = ctor(pipeline=pipeline)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        for name, m in model.named_modules():
            if name.endswith('conv_offset') or name.endswith('conv2_offset'):
                m.weight.copy_(torch.randn(m.weight.shape, generator=g) * offset_std)
        if cls_gain is not None:
            model.rcnn.cls_score.weight.mul_(cls_gain)
    model = model.to(device)
    model.prepare_inference(channels_last=channels_last, fold_bn=fold_bn)
    return model
