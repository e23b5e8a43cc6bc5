"""Synthetic inputs and seeded weights for the benchmark / tests (no datasets or checkpoints offline).

Image: integer-valued U{0..255} BGR minus config.network.pixel_means, fp32 NCHW,
zero-padded to a multiple of 32 (base_dataset.py:910-920), im_info = [[H, W, 1.0]] with the unpadded size.

Weights: the modules' own initialisers under torch.manual_seed(235) (the reference's seed,
upsnet_end2end_train.py:67-69) plus three documented, seeded adjustments that make the random network
behave like a trained one where that matters for the hot path:
  (1) frozen-BN statistics: the reference's checkpoints carry running_mean / running_var of a trained
      backbone; an untrained identity BN (mean 0, var 1) lets the +-128 pixel scale run through all 50 layers
      (activations of magnitude 50-140, where an absolute 1e-4 on fp32 logits means nothing). `calibrate_statistics`
      sets every frozen BN's running statistics from ONE seeded calibration image pushed through the
      layers in order (a data-dependent initialisation: each BN then emits zero-mean / unit-variance channels
      on that image), so activations are O(1-10) everywhere as in a trained network;
  (2) deformable offset convolutions get N(0, s) weights with s chosen so that the predicted offsets have
      a standard deviation of `offset_px` pixels on the calibration image (default 1 px: trained-model-like;
      the reference zero-initialises them, which would make every DCN a plain conv);
  (3) rcnn.cls_score is rescaled (`cls_gain`) so that a realistic number of detections survive the 0.6
      panoptic threshold. Every benchmark number is reported together with n_rois / n_det / n_inst.
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

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


# classifier gain per class count (tools/calib_gain_cpu.py, host-only sweep): ~100 detections and 40-100 panoptic detections
# above the 0.6 threshold, no saturated (tied) probabilities
DEFAULT_CLS_GAIN = {9: 5.4, 81: 22.0}
# standard deviation of the predicted sampling offsets on the calibration image: 1 px, i.e. |offset| <= ~5 px over a whole image, like a
# trained DCN (the op-level tests and microbenchmarks use the N(0, 2^2) px of SURVEY 8d; build_model(offset_px=...) widens the model's)
DEFAULT_OFFSET_PX = 1.0
BACKBONE_OFFSET_PX = 1.0


def _set_bn(bn, y, gamma=1.0):
    """running statistics of a frozen BN := per-channel statistics of its input y on the calibration image; weight := gamma."""
    bn.weight.fill_(gamma)
    bn.running_mean.copy_(y.mean(dim=(0, 2, 3)))
    bn.running_var.copy_(y.var(dim=(0, 2, 3), unbiased=False).clamp_min(1e-6))


def _scale_offset_conv(conv, x, g, offset_px):
    """N(0,1) weights rescaled so that conv(x) (the predicted sampling offsets) has a standard deviation of offset_px."""
    conv.weight.copy_(torch.randn(conv.weight.shape, generator=g))
    conv.bias.zero_()
    std = float(F.conv2d(x, conv.weight, None, conv.stride, conv.padding).std())
    conv.weight.mul_(offset_px / max(std, 1e-12))


def calibrate_statistics(model, seed, offset_px=DEFAULT_OFFSET_PX, size=(128, 256), gamma=0.6, gamma_last=0.2):
    """Data-dependent, seeded initialisation of everything a checkpoint would supply beyond the initialisers' scale (module
    docstring, items 1 and 2). Runs on the CPU in fp32 with plain torch calls, layer by layer in graph order (resnet.py:53-175,
    347-356; fpn.py:78-104; fcn.py:29-58), before BN folding. Deformable 3x3 layers are evaluated at zero offsets here (only
    their output statistics are needed). BN weights: gamma on bn1 / bn2 / stem / projection, gamma_last on the last BN of each
    bottleneck (trained ResNets carry small weights there: the residual branch is a correction to the shortcut, not its equal).
    With gamma = gamma_last = 1 the random network is an EXPANDING map -- the rounding noise of any fp32 execution grows ~4x per
    stage (measured: torch-CPU fp32 vs float64, 5e-4 at res5 on values of magnitude 8), which no trained network does; with
    0.6 / 0.2 it is mildly contracting like the uncalibrated initialisation, at activations of magnitude 1-10."""
    import torch.nn as nn
    g = torch.Generator().manual_seed(seed + 1)
    x = make_image(size[0], size[1], seed=seed + 2)['data']

    def conv_bn(x, conv, bn, relu, gamma=gamma):
        y = F.conv2d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation)
        if isinstance(bn, nn.BatchNorm2d):
            _set_bn(bn, y, gamma)
            y = F.batch_norm(y, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.0, bn.eps)
        return F.relu(y) if relu else y

    with torch.no_grad():
        bb = model.resnet_backbone
        y = F.max_pool2d(conv_bn(x, bb.conv1.conv1, bb.conv1.bn1, True), 3, 2, 1)
        feats = []
        for name in ('res2', 'res3', 'res4', 'res5'):
            for blk in getattr(bb, name).layers:
                t = conv_bn(y, blk.conv1, blk.bn1, True)
                if hasattr(blk, 'conv2_offset'):
                    _scale_offset_conv(blk.conv2_offset, t, g, min(offset_px, BACKBONE_OFFSET_PX))
                t = conv_bn(t, blk.conv2, blk.bn2, True)
                t = conv_bn(t, blk.conv3, blk.bn3, False, gamma_last)
                sc = y if blk.downsample is None else conv_bn(y, blk.downsample[0], blk.downsample[1], False)
                y = F.relu(t + sc)
            feats.append(y)
        pyr = model.fpn(*feats)     # (plain library convolutions on CPU tensors)
        for lvl in pyr[:1]:        # the subnet is shared by P2..P5: its offset predictors are scaled on the largest map
            t = lvl
            for i in range(model.fcn_head.fcn_subnet.num_layers):
                layer = model.fcn_head.fcn_subnet.conv[i][0]
                _scale_offset_conv(layer.conv_offset, t, g, offset_px)
                t = F.relu(F.conv2d(t, layer.conv.weight, layer.conv.bias, layer.conv.stride, layer.conv.padding, layer.conv.dilation))


def build_unprepared(symbol=None, seed=235, offset_px=DEFAULT_OFFSET_PX, cls_gain='default', pipeline='fused', calibrate=True, **calib_kw):
    """The seeded synthetic model as a CHECKPOINT would hold it: on the CPU, frozen BNs unfolded, the reference's state-dict keys
    (resnet.py:221-299). `state_dict()` of this object is what tests save as a stand-in for a trained .pth; build_model() =
    build_unprepared() -> .to(device) -> prepare_inference()."""
    from .models.resnet_upsnet import resnet_50_upsnet, resnet_101_upsnet
    ctor = {'resnet_50_upsnet': resnet_50_upsnet, 'resnet_101_upsnet': resnet_101_upsnet}[symbol or config.symbol]
    torch.manual_seed(seed)
    with torch.device('cpu'):
        model = ctor(pipeline=pipeline)
    model = model.to('cpu')
    with torch.no_grad():
        if calibrate:
            calibrate_statistics(model, seed, offset_px=offset_px, **calib_kw)
        else:
            g = torch.Generator().manual_seed(seed + 1)
            for name, m in model.named_modules():
                if name.endswith('conv_offset') or name.endswith('conv2_offset'):
                    m.weight.copy_(torch.randn(m.weight.shape, generator=g) * 0.01)
        if cls_gain == 'default':
            cls_gain = DEFAULT_CLS_GAIN.get(config.dataset.num_classes, 5.4) if calibrate else 0.3
        if cls_gain is not None:
            model.rcnn.cls_score.weight.mul_(cls_gain)
    return model


def build_model(symbol=None, seed=235, device='cuda', offset_px=DEFAULT_OFFSET_PX, cls_gain='default', pipeline='fused',
                channels_last=True, fold_bn=True, calibrate=True, **calib_kw):
    """Construct the configured model with seeded synthetic weights, ready for inference.
    calibrate=False reproduces the r01-r07 synthetic model (identity BN, offset weights N(0, 0.01): activations of magnitude
    50-140 and offsets of up to +-26 px -- kept for A/B runs of the deformable kernels on wide offsets)."""
    model = build_unprepared(symbol, seed, offset_px, cls_gain, pipeline, calibrate, **calib_kw)
    model = model.to(device)
    model.prepare_inference(channels_last=channels_last, fold_bn=fold_bn)
    return model
