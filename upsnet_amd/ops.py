"""Functional wrappers over the C ABI (tensor plumbing only: allocation, layout views, stream).

Every function here launches hand-written HIP kernels from libupsnet_hip.so on the current torch
stream. Nothing falls back to PyTorch/CPU; non-CUDA inputs raise (as the reference's Functions do,
functions/deform_conv.py:40-41, functions/roialign.py:34-35).
"""
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, f32c, float_array, int_array, lib, nhwc, ptr, ptr_array, require_cuda, stream


# bench.py sets PROFILE['enabled'] to bracket the dominant kernel with events on the launch stream
PROFILE = {'enabled': False, 'events': []}


def last_kernel_form():
    """Test introspection (include/upsnet_hip.h: upsnet_last_kernel_form): the kernel instance this thread's last library-dispatched
    convolution launch took."""
    return lib().upsnet_last_kernel_form().decode()


def _ws(nbytes, device):
    return torch.empty((int(nbytes),), dtype=torch.uint8, device=device)


# ----------------------------------------------------------------------------- ROIAlign
def roi_align_nchw(features, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio=2):
    """Drop-in for roi_align_cuda.roi_align_forward (NCHW in, [N,C,PH,PW] out)."""
    require_cuda(features, rois)
    features, rois = f32c(features), f32c(rois)
    if rois.dim() != 2 or rois.shape[1] != 5:
        raise RuntimeError("rois must be [N,5]")
    B, C, H, W = features.shape
    N = rois.shape[0]
    out = torch.empty((N, C, pooled_h, pooled_w), dtype=torch.float32, device=features.device)
    check(lib().upsnet_roi_align_forward(stream(), ptr(features), float(spatial_scale), N, H, W, C, int(pooled_h),
                                         int(pooled_w), int(sampling_ratio), ptr(rois), ptr(out)), "roi_align_forward")
    return out


def roi_align_nhwc(features, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio=2):
    """ROIAlign on channels-last features; returns a channels_last [N,C,PH,PW] tensor."""
    require_cuda(features, rois)
    features, rois = nhwc(features.float()), f32c(rois)
    B, C, H, W = features.shape
    N = rois.shape[0]
    out = torch.empty((N, C, pooled_h, pooled_w), dtype=torch.float32, device=features.device,
                      memory_format=torch.channels_last)
    check(lib().upsnet_roi_align_forward_nhwc(stream(), ptr(features), B, H, W, C, float(spatial_scale), ptr(rois), N,
                                              int(pooled_h), int(pooled_w), int(sampling_ratio), ptr(out)),
          "roi_align_forward_nhwc")
    return out


def roi_align_algorithmic_bytes(feats, n, C, ph, pw):
    """the pooled output written once + the ROI records + the feature pixels the ROIs can touch read once (per ROI the
    (2 ph + 1) x (2 pw + 1) sample neighbourhood of its level, capped at the whole pyramid)."""
    feat_bytes = 4 * C * sum(f.shape[2] * f.shape[3] for f in feats)
    return 4.0 * n * C * ph * pw + 20.0 * n + min(feat_bytes, 4.0 * n * C * (2 * ph + 1) * (2 * pw + 1))


# r13: deal the ROIs to the XCDs by image neighbourhood (csrc/roi_order.h). 'auto': launches of >= ROI_XCD_ORDER_MIN ROIs whose rois tensor carries
# the table pyramid_proposals wrote for it (the box head's 1000 proposals). Results are bit-identical either way.
ROI_XCD_ORDER = os.environ.get('UPSNET_ROI_XCD_ORDER', '1') != '0'
ROI_XCD_ORDER_MIN = int(os.environ.get('UPSNET_ROI_XCD_ORDER_MIN', '512'))
# a ROI set that does not come with a table (not straight from pyramid_proposals) gets one from a launch of its own only if this is set: measured
# (profiles/r13_bench.log, random ROIs) the extra launch costs what the dealing saves cold -- 1000 ROIs 63.6 vs 62.7 us, 300 ROIs 28.3 vs 25.2
ROI_XCD_ORDER_STANDALONE = os.environ.get('UPSNET_ROI_XCD_ORDER_STANDALONE', '0') != '0'


def fpn_roi_order(rois, image_hw, num_rois_dev=None):
    """int32 [N]: the workgroup -> ROI table of fpn_roi_align(order=...) that gives every XCD one contiguous range of the ROIs sorted by
    (pyramid level, image stripe, column cell). image_hw: extent of the image the ROIs live in. N <= 2048."""
    require_cuda(rois)
    rois = f32c(rois)
    order = torch.empty((max(rois.shape[0], 1),), dtype=torch.int32, device=rois.device)
    check(lib().upsnet_fpn_roi_order(stream(), ptr(rois), int(rois.shape[0]), ptr(num_rois_dev), int(image_hw[0]), int(image_hw[1]), ptr(order)),
          "fpn_roi_order")
    return order[:rois.shape[0]]


def fpn_roi_align(feats, rois, pooled_h, pooled_w, spatial_scale, sampling_ratio=2, num_rois_dev=None,
                  return_levels=False, order='auto'):
    """FPNRoIAlign.forward on device: feats = 4 logical-NCHW tensors (batch 1), rois [N,5].
    order: None = workgroup b takes ROI b; an int32 [N] table from fpn_roi_order; 'auto' = build the table when the launch has at least
    ROI_XCD_ORDER_MIN ROIs (UPSNET_ROI_XCD_ORDER=0 switches it off). Same bits in every case.

    Returns a channels_last [N,C,PH,PW] tensor in the original ROI order."""
    require_cuda(rois, *feats)
    assert len(feats) == 4 and len(spatial_scale) == 4
    feats = [nhwc(f.float()) for f in feats]
    rois = f32c(rois)
    C = feats[0].shape[1]
    for f in feats:
        if f.shape[0] != 1 or f.shape[1] != C:
            raise RuntimeError("fpn_roi_align: batch 1 and equal channel counts required")
    N = rois.shape[0]
    dev = rois.device
    out = torch.empty((N, C, pooled_h, pooled_w), dtype=torch.float32, device=dev, memory_format=torch.channels_last)
    levels = torch.empty((max(N, 1),), dtype=torch.int32, device=dev) if return_levels else None
    if isinstance(order, str):
        given = getattr(rois, '_ups_roi_order', None)          # the proposals' own table (pyramid_proposals wrote it: no launch here)
        eligible = ROI_XCD_ORDER and sampling_ratio == 2 and ROI_XCD_ORDER_MIN <= N <= 2048 and max(pooled_h, pooled_w) <= 32
        if eligible and given is not None and given.numel() == N and given.device == rois.device:
            order = given
        elif eligible and ROI_XCD_ORDER_STANDALONE:
            order = fpn_roi_order(rois, (int(round(feats[0].shape[2] / spatial_scale[0])), int(round(feats[0].shape[3] / spatial_scale[0]))), num_rois_dev)
        else:
            order = None
    elif order is not None:
        # (the kernel trusts the table: a table that is not a permutation of 0..N-1 would leave output rows unwritten)
        if not (isinstance(order, torch.Tensor) and order.dtype == torch.int32 and order.numel() == N and order.device == rois.device and order.is_contiguous()):
            raise RuntimeError("fpn_roi_align: order must be a contiguous int32 tensor of %d entries on %s (from fpn_roi_order / pyramid_proposals)" % (N, rois.device))
    if PROFILE['enabled']:
        ev0 = torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_fpn_roi_align_forward_ordered(stream(), ptr_array(feats), int_array([f.shape[2] for f in feats]),
                                                     int_array([f.shape[3] for f in feats]), float_array(spatial_scale), C,
                                                     ptr(rois), N, ptr(num_rois_dev), int(pooled_h), int(pooled_w),
                                                     int(sampling_ratio), ptr(out), ptr(levels), ptr(order)), "fpn_roi_align_forward_ordered")
    if PROFILE['enabled']:
        ev1 = torch.cuda.Event(enable_timing=True)
        ev1.record()
        PROFILE['events'].append(('roialign', ev0, ev1, 0.0, roi_align_algorithmic_bytes(feats, N, C, pooled_h, pooled_w),
                                  'roialign %dx%dx%dx%d' % (N, C, pooled_h, pooled_w)))
    return (out, levels[:N]) if return_levels else out


# ----------------------------------------------------------------------------- deformable conv
def out_hw(H, W, k, pad, stride, dil):
    return ((H + 2 * pad[0] - dil[0] * (k[0] - 1) - 1) // stride[0] + 1,
            (W + 2 * pad[1] - dil[1] * (k[1] - 1) - 1) // stride[1] + 1)


def deform_im2col(data_im, data_offset, im_shape, col_shape, kernel_shape, pad, stride, dilation, parallel_imgs,
                  deformable_group, data_col):
    """Drop-in for deform_conv_cuda.deform_im2col (deform_conv_cuda.cpp:49-68); fills data_col in place."""
    require_cuda(data_im, data_offset, data_col)
    assert data_im.is_contiguous() and data_offset.is_contiguous() and data_col.is_contiguous()
    check(lib().upsnet_deform_im2col(stream(), ptr(data_im), ptr(data_offset), int(im_shape[1]), int(im_shape[2]),
                                     int(im_shape[3]), int(kernel_shape[0]), int(kernel_shape[1]), int(pad[0]), int(pad[1]),
                                     int(stride[0]), int(stride[1]), int(dilation[0]), int(dilation[1]), int(parallel_imgs),
                                     int(deformable_group), ptr(data_col)), "deform_im2col")
    return 1


def mod_deform_im2col(data_im, data_offset, data_mask, im_shape, col_shape, kernel_shape, pad, stride, dilation,
                      deformable_group, data_col):
    """Drop-in for mod_deform_conv_cuda.mod_deform_im2col (mod_deform_conv_cuda.cpp:51-74), batch 1."""
    require_cuda(data_im, data_offset, data_mask, data_col)
    assert data_im.is_contiguous() and data_offset.is_contiguous() and data_mask.is_contiguous()
    check(lib().upsnet_mod_deform_im2col(stream(), ptr(data_im), ptr(data_offset), ptr(data_mask), 1, int(im_shape[1]),
                                         int(im_shape[2]), int(im_shape[3]), int(col_shape[1]), int(col_shape[2]),
                                         int(kernel_shape[0]), int(kernel_shape[1]), int(pad[0]), int(pad[1]),
                                         int(stride[0]), int(stride[1]), int(dilation[0]), int(dilation[1]),
                                         int(deformable_group), ptr(data_col)), "mod_deform_im2col")
    return 1


def _shape_args(im_shape, kernel_shape, pad, stride, dilation):
    return (int(im_shape[1]), int(im_shape[2]), int(im_shape[3]), int(kernel_shape[0]), int(kernel_shape[1]), int(pad[0]),
            int(pad[1]), int(stride[0]), int(stride[1]), int(dilation[0]), int(dilation[1]))


def deform_col2im(data_col, data_offset, im_shape, col_shape, kernel_shape, pad, stride, dilation, parallel_imgs,
                  deformable_group, grad_im):
    """Drop-in for deform_conv_cuda.deform_col2im (deform_conv_cuda.cpp:70-86); accumulates into grad_im."""
    require_cuda(data_col, data_offset, grad_im)
    assert data_col.is_contiguous() and data_offset.is_contiguous() and grad_im.is_contiguous()
    check(lib().upsnet_deform_col2im(stream(), ptr(data_col), ptr(data_offset), *_shape_args(im_shape, kernel_shape, pad, stride,
                                                                                            dilation),
                                     int(parallel_imgs), int(deformable_group), ptr(grad_im)), "deform_col2im")
    return 1


def deform_col2im_coord(data_col, data_im, data_offset, im_shape, col_shape, kernel_shape, pad, stride, dilation,
                        parallel_imgs, deformable_group, grad_offset):
    """Drop-in for deform_conv_cuda.deform_col2im_coord (deform_conv_cuda.cpp:88-105); overwrites grad_offset."""
    require_cuda(data_col, data_im, data_offset, grad_offset)
    assert data_col.is_contiguous() and data_im.is_contiguous() and data_offset.is_contiguous() and grad_offset.is_contiguous()
    check(lib().upsnet_deform_col2im_coord(stream(), ptr(data_col), ptr(data_im), ptr(data_offset),
                                           *_shape_args(im_shape, kernel_shape, pad, stride, dilation), int(parallel_imgs),
                                           int(deformable_group), ptr(grad_offset)), "deform_col2im_coord")
    return 1


def mod_deform_col2im(data_col, data_offset, data_mask, im_shape, col_shape, kernel_shape, pad, stride, dilation,
                      deformable_group, grad_im):
    """Drop-in for mod_deform_conv_cuda.mod_deform_col2im (mod_deform_conv_cuda.cpp:76-92), batch 1."""
    require_cuda(data_col, data_offset, data_mask, grad_im)
    assert data_col.is_contiguous() and data_offset.is_contiguous() and data_mask.is_contiguous() and grad_im.is_contiguous()
    a = _shape_args(im_shape, kernel_shape, pad, stride, dilation)
    check(lib().upsnet_mod_deform_col2im(stream(), ptr(data_col), ptr(data_offset), ptr(data_mask), 1, a[0], a[1], a[2],
                                         int(col_shape[1]), int(col_shape[2]), *a[3:], int(deformable_group), ptr(grad_im)),
          "mod_deform_col2im")
    return 1


def mod_deform_col2im_coord(data_col, data_im, data_offset, data_mask, im_shape, col_shape, kernel_shape, pad, stride,
                            dilation, deformable_group, grad_offset, grad_mask):
    """Drop-in for mod_deform_conv_cuda.mod_deform_col2im_coord (mod_deform_conv_cuda.cpp:94-114), batch 1."""
    require_cuda(data_col, data_im, data_offset, data_mask, grad_offset, grad_mask)
    assert data_col.is_contiguous() and data_im.is_contiguous() and data_offset.is_contiguous() and data_mask.is_contiguous()
    assert grad_offset.is_contiguous() and grad_mask.is_contiguous()
    a = _shape_args(im_shape, kernel_shape, pad, stride, dilation)
    check(lib().upsnet_mod_deform_col2im_coord(stream(), ptr(data_col), ptr(data_im), ptr(data_offset), ptr(data_mask), 1, a[0],
                                               a[1], a[2], int(col_shape[1]), int(col_shape[2]), *a[3:], int(deformable_group),
                                               ptr(grad_offset), ptr(grad_mask)), "mod_deform_col2im_coord")
    return 1


def roi_align_backward(pooled_h, pooled_w, sampling_ratio, spatial_scale, top_grad, rois, bottom_grad):
    """Drop-in for roi_align_cuda.roi_align_backward (roi_align_cuda.cpp:77-112); accumulates into bottom_grad [B,C,H,W]."""
    require_cuda(top_grad, rois, bottom_grad)
    if rois.dim() != 2 or rois.shape[1] != 5:
        return 0
    assert top_grad.is_contiguous() and rois.is_contiguous() and bottom_grad.is_contiguous()
    assert top_grad.dtype == torch.float32 and rois.dtype == torch.float32 and bottom_grad.dtype == torch.float32
    B, C, H, W = bottom_grad.shape
    check(lib().upsnet_roi_align_backward(stream(), ptr(top_grad), float(spatial_scale), B, rois.shape[0], H, W, C, int(pooled_h),
                                          int(pooled_w), int(sampling_ratio), ptr(rois), ptr(bottom_grad)), "roi_align_backward")
    return 1


# Deformable-convolution kernel generation: 'frag' = csrc/deform_fused.hip (default), 'igemm' = the first-generation loader mode of
# the dense kernel (csrc/conv.hip), kept for A/B measurements and for geometries the new kernel does not take.
DCN_KERNEL = os.environ.get('UPSNET_DCN_KERNEL', 'frag')


def dcn_precision():
    """'bf16' when the dense convolutions run on the bf16 matrix cores (hipconv.PRECISION == 'bf16', BASELINE configs[2]): the fused
    deformable convolutions then do too (csrc/deform_fused_bf16.hip). 'bf16x3' and 'fp32' keep the exact fp32 kernel."""
    from .models import hipconv
    return 'bf16' if hipconv.PRECISION == 'bf16' and os.environ.get('UPSNET_DCN_BF16', '1') != '0' else 'fp32'


def dcn_square(pad, stride, dil):
    """The fused kernels (both generations) take one pad / stride / dilation for both axes; other geometries run the reference's own
    structure (im2col into a caller-allocated column buffer + one GEMM) through the NCHW drop-in entry point."""
    return pad[0] == pad[1] and stride[0] == stride[1] and dil[0] == dil[1]


def pack_dcn_weight(weight, kind=None):
    """[Cout,Cin,kh,kw] -> packed weight for deform_conv_fused: ('frag', wp) in MFMA fragment order for csrc/deform_fused.hip
    (('frag_bf16', wp) for csrc/deform_fused_bf16.hip: kind 'frag_bf16', or 'frag' while dcn_precision() is 'bf16'), or
    (wpack [kh*kw*Cin, ldw], ldw) -- the dense convolution's packing -- for the first-generation kernel."""
    kind = kind or DCN_KERNEL
    cout, cin, kh, kw = weight.shape
    if kind == 'frag' and dcn_precision() == 'bf16':
        kind = 'frag_bf16'
    if kind == 'frag_bf16' and cin % 32 == 0 and kh * kw <= 25:
        w = f32c(weight)
        wp = torch.empty((lib().upsnet_dcn_packed_weight_bf16_elems(cout, cin, kh, kw),), dtype=torch.bfloat16, device=w.device)
        check(lib().upsnet_dcn_pack_weight_bf16(stream(), ptr(w), cout, cin, kh, kw, ptr(wp)), "dcn_pack_weight_bf16")
        return ('frag_bf16', wp)
    if kind == 'frag' and cin % 32 == 0 and kh * kw <= 25:
        w = f32c(weight)
        wp = torch.empty((lib().upsnet_dcn_packed_weight_floats(cout, cin, kh, kw),), dtype=torch.float32, device=w.device)
        check(lib().upsnet_dcn_pack_weight(stream(), ptr(w), cout, cin, kh, kw, ptr(wp)), "dcn_pack_weight")
        return ('frag', wp)
    return pack_conv_weight(weight)


def cached_dcn_pack(weight):
    """Packed deformable-convolution weight cached ON the weight tensor (re-packed when it changes or moves)."""
    key = (weight.data_ptr(), weight._version, tuple(weight.shape), DCN_KERNEL, dcn_precision())
    ent = weight.__dict__.get('_ups_dcn_pack')
    if ent is None or ent[0] != key:
        ent = (key, pack_dcn_weight(weight.detach()))
        weight.__dict__['_ups_dcn_pack'] = ent
    return ent[1]


def fused_dcn_supported(cin, cout, deformable_groups, groups, pad=(0, 0), stride=(1, 1), dil=(1, 1)):
    return groups == 1 and deformable_groups == 1 and cin % 32 == 0 and dcn_square(pad, stride, dil)


def _nhwc_out(n, c, h, w, device):
    """channels_last [n,c,h,w] output whose memory is NHWC-dense even for degenerate dims."""
    return torch.empty((n, h, w, c), dtype=torch.float32, device=device).permute(0, 3, 1, 2)


DCN_SPLITK = os.environ.get('UPSNET_DCN_SPLITK', '1') != '0'


def dcn_ksplit(outs, cin, cout, taps):
    """Split-K factor of the fused deformable kernel for one small map: enough workgroups for ~2 per CU (~3 for the smallest maps), >= 9 K steps each."""
    if not DCN_SPLITK or cout % 4:
        return 1
    o = outs[0]
    wgs = o.shape[0] * -(-o.shape[2] // 8) * -(-o.shape[3] // 8) * -(-cout // 128)
    nsl = cin // 32 * taps
    # (r13) maps of <= 128 tiles (res5 of UPSNet-101-DCN: 96 at 800x1333, 128 at 1024x2048) split until ~3 workgroups per CU: 512 -> 512 on 25 x 42
    # x4 86.8, x6 84.3, x8 68.6 us (tools/bench_c3_layers.py); larger maps are flat beyond 384 workgroups (256 -> 256 on 50 x 84: x3 64.0, x4 68.3, x6 64.0)
    target = 768 if wgs <= 128 else 384
    ks = 1
    while ks < 8 and wgs * ks < target and nsl // (ks + 1) >= 9:
        ks += 1
    while ks > 1 and -(-nsl // ks) * (ks - 1) >= nsl:
        ks -= 1
    return ks


def _dcn_event(ev0, xs, outs, cin, cout, ksize, kind='dcn_fused'):
    ev1 = torch.cuda.Event(enable_timing=True)
    ev1.record()
    npix = sum(o.shape[2] * o.shape[3] for o in outs)
    taps = ksize[0] * ksize[1]
    PROFILE['events'].append((kind, ev0, ev1, 2.0 * cout * cin * taps * npix,
                              4.0 * (sum(x.shape[2] * x.shape[3] for x in xs) * cin + npix * (2 * taps + cout) + cout * cin * taps),
                              "dcn %d->%d %s" % (cin, cout, [tuple(x.shape[2:]) for x in xs])))


def deform_conv_fused(xs, offsets, wpack, bias, cin, cout, ksize, stride, pad, dil, masks=None, relu=False):
    """Fused deformable conv over up to 4 maps sharing weights. xs/offsets/masks: lists of logical NCHW
    tensors with batch 1; wpack = (packed weight, ldw) from pack_dcn_weight; returns channels_last outputs."""
    wp, ldw = wpack     # ('frag', packed) or (packed, ldw)
    require_cuda(ldw if isinstance(wp, str) else wp, *xs)
    n = len(xs)
    assert 1 <= n <= 4 and len(offsets) == n
    xs = [nhwc(x.float()) for x in xs]
    offsets = [nhwc(o.float()) for o in offsets]
    if masks is not None:
        masks = [nhwc(m.float()) for m in masks]
    outs = []
    for x, o in zip(xs, offsets):
        if x.shape[0] != 1 or x.shape[1] != cin:
            raise RuntimeError("deform_conv_fused: batch 1 / Cin mismatch")
        Ho, Wo = out_hw(x.shape[2], x.shape[3], ksize, pad, stride, dil)
        if tuple(o.shape) != (1, 2 * ksize[0] * ksize[1], Ho, Wo):
            raise RuntimeError("deform_conv_fused: offset shape %s != %s" % (tuple(o.shape), (1, 2 * ksize[0] * ksize[1], Ho, Wo)))
        outs.append(_nhwc_out(1, cout, Ho, Wo, x.device))
    b = None if bias is None else f32c(bias)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if wp == 'frag_bf16':   # bf16 matrix cores (BASELINE configs[2])
        if not (pad[0] == pad[1] and stride[0] == stride[1] and dil[0] == dil[1]):
            raise RuntimeError("deform_conv_fused: square pad / stride / dilation only")
        check(lib().upsnet_deform_conv_fused_nhwc_bf16(stream(), n, ptr_array(xs), ptr_array(offsets),
                                                       ptr_array(masks) if masks is not None else None, ptr_array(outs),
                                                       int_array([x.shape[2] for x in xs]), int_array([x.shape[3] for x in xs]),
                                                       int(cin), int(cout), ksize[0], ksize[1], pad[0], stride[0], dil[0], ptr(ldw), ptr(b),
                                                       int(bool(relu))), "deform_conv_fused_nhwc_bf16")
        if PROFILE['enabled']:
            _dcn_event(ev0, xs, outs, cin, cout, ksize, 'dcn_fused_bf16')
        return outs
    if wp == 'frag':   # (kind, packed) from pack_dcn_weight: second-generation kernel
        if not (pad[0] == pad[1] and stride[0] == stride[1] and dil[0] == dil[1]):
            raise RuntimeError("deform_conv_fused: square pad / stride / dilation only")
        ks = dcn_ksplit(outs, cin, cout, ksize[0] * ksize[1]) if n == 1 else 1
        if ks > 1:
            wsb = lib().upsnet_deform_conv_fused_splitk_workspace_bytes(xs[0].shape[2], xs[0].shape[3], int(cout), ksize[0], ksize[1], pad[0],
                                                                        stride[0], dil[0], ks)
            ws = _ws(wsb, xs[0].device)
            check(lib().upsnet_deform_conv_fused_nhwc_splitk(stream(), ptr(xs[0]), ptr(offsets[0]), ptr(masks[0]) if masks is not None else None,
                                                             ptr(outs[0]), xs[0].shape[2], xs[0].shape[3], int(cin), int(cout), ksize[0],
                                                             ksize[1], pad[0], stride[0], dil[0], ptr(ldw), ptr(b), int(bool(relu)), ks, ptr(ws)),
                  "deform_conv_fused_nhwc_splitk")
            if PROFILE['enabled']:
                _dcn_event(ev0, xs, outs, cin, cout, ksize)
            return outs
        check(lib().upsnet_deform_conv_fused_nhwc(stream(), n, ptr_array(xs), ptr_array(offsets),
                                                  ptr_array(masks) if masks is not None else None, ptr_array(outs),
                                                  int_array([x.shape[2] for x in xs]), int_array([x.shape[3] for x in xs]),
                                                  int(cin), int(cout), ksize[0], ksize[1], pad[0], stride[0], dil[0], ptr(ldw), ptr(b),
                                                  int(bool(relu))), "deform_conv_fused_nhwc")
    else:
        check(lib().upsnet_deform_conv_forward_nhwc(stream(), n, ptr_array(xs), ptr_array(offsets),
                                                    ptr_array(masks) if masks is not None else None, ptr_array(outs),
                                                    int_array([x.shape[2] for x in xs]), int_array([x.shape[3] for x in xs]),
                                                    int(cin), int(cout), ksize[0], ksize[1], pad[0], pad[1], stride[0], stride[1],
                                                    dil[0], dil[1], 1, ptr(wp), int(ldw), ptr(b), int(bool(relu))),
              "deform_conv_forward_nhwc")
    if PROFILE['enabled']:
        _dcn_event(ev0, xs, outs, cin, cout, ksize)
    return outs


# ----------------------------------------------------------------------------- NMS
def nms_host(sorted_dets, thresh, device_id=0):
    """`_nms` drop-in: numpy float32 [N,>=4] sorted by score -> kept indices (numpy int32)."""
    d = np.ascontiguousarray(sorted_dets, dtype=np.float32)
    n = d.shape[0]
    keep = np.zeros((max(n, 1),), np.int32)
    num = np.zeros((1,), np.int32)
    check(lib().upsnet_nms_host(keep.ctypes.data_as(_lib.c_void_p), num.ctypes.data_as(_lib.c_void_p),
                                d.ctypes.data_as(_lib.c_void_p), n, d.shape[1] if d.ndim == 2 else 5, float(thresh),
                                int(device_id)), "nms_host")
    return keep[:int(num[0])]


def nms_batched(boxes, scores, counts, thresh, pre_removed=None):
    """boxes [P,nmax,4], scores [P,nmax], counts [P] int32 (device) -> keep_idx [P,nmax] int32, keep_cnt [P]."""
    require_cuda(boxes, scores, counts)
    boxes, scores = f32c(boxes), f32c(scores)
    counts = counts.to(torch.int32).contiguous()
    Pn, nmax = scores.shape
    dev = boxes.device
    keep = torch.empty((Pn, nmax), dtype=torch.int32, device=dev)
    cnt = torch.empty((Pn,), dtype=torch.int32, device=dev)
    ws = _ws(lib().upsnet_nms_workspace_bytes(Pn, nmax), dev)
    pr = None if pre_removed is None else pre_removed.to(torch.uint8).contiguous()
    check(lib().upsnet_nms_batched(stream(), ptr(boxes), ptr(scores), ptr(counts), ptr(pr), Pn, nmax, float(thresh),
                                   ptr(keep), ptr(cnt), ptr(ws)), "nms_batched")
    return keep, cnt


def gpu_nms(dets, thresh):
    """gpu_nms (gpu_nms.pyx:23-38) on a device tensor [N,5]; returns a device int64 tensor of kept indices
    (visiting order). One tiny D2H (the count) to size the result."""
    require_cuda(dets)
    dets = f32c(dets)
    n = dets.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dets.device)
    counts = torch.full((1,), n, dtype=torch.int32, device=dets.device)
    keep, cnt = nms_batched(dets[None, :, :4], dets[None, :, 4], counts, thresh)
    return keep[0, :int(cnt.item())].long()


def cpu_nms(dets, thresh):
    """cpu_nms (cpu_nms.pyx:29-80) on a device tensor [N,5]: like gpu_nms but suppression at `overlap >= thresh` (double compare)."""
    require_cuda(dets)
    dets = f32c(dets)
    n = dets.shape[0]
    if n == 0:
        return torch.zeros((0,), dtype=torch.int64, device=dets.device)
    counts = torch.full((1,), n, dtype=torch.int32, device=dets.device)
    boxes, scores = f32c(dets[None, :, :4]), f32c(dets[None, :, 4])
    keep = torch.empty((1, n), dtype=torch.int32, device=dets.device)
    cnt = torch.empty((1,), dtype=torch.int32, device=dets.device)
    ws = _ws(lib().upsnet_nms_workspace_bytes(1, n), dets.device)
    check(lib().upsnet_cpu_nms_batched(stream(), ptr(boxes), ptr(scores), ptr(counts), 1, n, float(thresh), ptr(keep), ptr(cnt),
                                       ptr(ws)), "cpu_nms_batched")
    return keep[0, :int(cnt.item())].long()


def soft_nms_batched(boxes, counts=None, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """P soft-NMS problems in one launch (one workgroup each): boxes [P,nmax,5] (x1,y1,x2,y2,score), counts [P] int32 device or
    None. Returns (boxes' [P,nmax,5], inds [P,nmax] int64, n_out [P] int32), all on the device: rows [0, n_out[p]) of problem p are
    cpu_soft_nms's result (cpu_nms.pyx:91-196), the rest is unspecified."""
    require_cuda(boxes)
    b = f32c(boxes).clone()
    Pn, nmax = b.shape[0], b.shape[1]
    inds = torch.empty((Pn, max(nmax, 1)), dtype=torch.int64, device=b.device)
    n_out = torch.zeros((max(Pn, 1),), dtype=torch.int32, device=b.device)
    if Pn and nmax:
        cnt = None if counts is None else counts.to(device=b.device, dtype=torch.int32).contiguous()
        ws = _ws(lib().upsnet_soft_nms_batched_workspace_bytes(Pn, nmax), b.device)
        check(lib().upsnet_soft_nms_batched(stream(), ptr(b), ptr(inds), ptr(cnt), Pn, nmax, float(sigma), float(Nt), float(threshold),
                                            int(method), ptr(n_out), ptr(ws)), "soft_nms_batched")
    return b, inds[:, :nmax], n_out[:Pn]


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """cpu_soft_nms semantics on device: returns (boxes', inds[:N'])."""
    require_cuda(boxes)
    b, inds, n_out = soft_nms_batched(f32c(boxes)[None], None, sigma, Nt, threshold, method)
    return b[0], inds[0, :int(n_out[0].item())] if boxes.shape[0] else inds[0, :0]


# ----------------------------------------------------------------------------- proposals
def pyramid_proposals(cls_probs, bbox_preds, im_info, anchors, strides, pre_nms_top_n, post_nms_top_n, nms_thresh,
                      min_size, joint=False):
    """Device-side PyramidProposal. cls_probs[l] [1,A,H,W], bbox_preds[l] [1,4A,H,W]; im_info device [3].
    Returns fixed-size (rois [post,5], scores [post], num device int32). joint = the individual_proposals=False branch
    (functions/pyramid_proposal.py:181-208): kept boxes in NMS order, before the reference's random padding."""
    require_cuda(im_info, *cls_probs)
    L = len(cls_probs)

    def pixel_linear(t):
        """(tensor, channel stride, pixel stride) if (c, h, w) -> c*cs + (h*W + w)*ps addresses t in place (NCHW, or a channel
        slice of an NHWC map), else a contiguous NCHW copy."""
        if t.dtype == torch.float32 and t.shape[0] == 1 and (t.shape[2] == 1 or t.stride(2) == t.shape[3] * t.stride(3)):
            return t, t.stride(1), t.stride(3)
        t = f32c(t)
        return t, t.stride(1), t.stride(3)
    cls_probs, cls_cs, cls_ps = zip(*[pixel_linear(c) for c in cls_probs])
    bbox_preds, box_cs, box_ps = zip(*[pixel_linear(b) for b in bbox_preds])
    A = cls_probs[0].shape[1]
    Hs = [c.shape[2] for c in cls_probs]
    Ws = [c.shape[3] for c in cls_probs]
    for c, b in zip(cls_probs, bbox_preds):
        if c.shape[0] != 1 or b.shape[1] != 4 * A:
            raise ValueError("Sorry, multiple images each device is not implemented")  # pyramid_proposal.py:47-49
    dev = im_info.device
    hs, ws_ = int_array(Hs), int_array(Ws)
    nbytes = lib().upsnet_proposal_workspace_bytes(L, hs, ws_, A, int(pre_nms_top_n), int(post_nms_top_n))
    if nbytes == 0:
        raise RuntimeError("pyramid_proposals: unsupported level count")
    ws = _ws(nbytes, dev)
    rois = torch.empty((post_nms_top_n, 5), dtype=torch.float32, device=dev)
    scores = torch.empty((post_nms_top_n,), dtype=torch.float32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    anc = float_array(np.asarray(anchors, np.float32).reshape(-1).tolist())
    long_array = lambda v: (_lib.c_long * len(v))(*[int(x) for x in v])
    args = (stream(), L, ptr_array(cls_probs), ptr_array(bbox_preds), long_array(cls_cs), long_array(cls_ps), long_array(box_cs), long_array(box_ps),
            hs, ws_, int_array(strides), anc, A, ptr(f32c(im_info)), int(pre_nms_top_n), int(post_nms_top_n), float(nms_thresh), float(min_size),
            ptr(rois), ptr(scores), ptr(num), ptr(ws))
    if joint:
        check(lib().upsnet_pyramid_proposals_joint_strided(*args), "pyramid_proposals")
    else:
        # r13: the launch that ranks the proposals also writes the ROI -> XCD dealing table of the box head's ROIAlign launch (csrc/roi_order.h);
        # it travels with the rois tensor (fpn_roi_align(order='auto') picks it up when it is handed this very tensor)
        order = torch.empty((post_nms_top_n,), dtype=torch.int32, device=dev) if (ROI_XCD_ORDER and ROI_XCD_ORDER_MIN <= post_nms_top_n <= 2048) else None
        check(lib().upsnet_pyramid_proposals_strided_ordered(*args, ptr(order)), "pyramid_proposals")
        if order is not None:
            rois._ups_roi_order = order
    return rois, scores, num


# ----------------------------------------------------------------------------- detection selection
def mask_roi(rois, bbox_delta, cls_prob, im_info, class_agnostic, score_thresh, nms_thresh, max_det, reg_weights,
             num_rois_dev=None, clip_boxes=True):
    """Device-side MaskROI: returns fixed-capacity (boxes [cap,5], scores [cap], cls [cap] int64, src [cap] int32,
    num device int32)."""
    require_cuda(rois, bbox_delta, cls_prob, im_info)
    rois, bbox_delta, cls_prob = f32c(rois), f32c(bbox_delta), f32c(cls_prob)
    N, C = cls_prob.shape
    dev = rois.device
    cap = lib().upsnet_mask_roi_capacity(N, C, int(class_agnostic))
    boxes = torch.empty((cap, 5), dtype=torch.float32, device=dev)
    scores = torch.empty((cap,), dtype=torch.float32, device=dev)
    cls = torch.empty((cap,), dtype=torch.int64, device=dev)
    src = torch.empty((cap,), dtype=torch.int32, device=dev)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _ws(lib().upsnet_mask_roi_workspace_bytes(N, C, int(class_agnostic)), dev)
    check(lib().upsnet_mask_roi_ex(stream(), ptr(rois), ptr(bbox_delta), ptr(cls_prob), N, ptr(num_rois_dev), C,
                                ptr(f32c(im_info.reshape(-1))), int(class_agnostic), int(bool(clip_boxes)), float(score_thresh), float(nms_thresh),
                                int(max_det), float_array(reg_weights), ptr(boxes), ptr(scores), ptr(cls), ptr(src),
                                ptr(num), ptr(ws)), "mask_roi")
    return boxes, scores, cls, src, num


def mask_logit_gather(mask_logit, row, cls, K=None):
    """pan_logit [K,1,ms,ms] with pan_logit[k] = mask_logit[row[k], cls[k]] (rows / classes clamped): one gather kernel instead of
    clamp + index_select + expand + gather. mask_logit: [n,C,ms,ms] in any dense layout (NCHW or channels_last)."""
    require_cuda(mask_logit, row, cls)
    n, C, mh, mw = mask_logit.shape
    t = mask_logit if mask_logit.dtype == torch.float32 else mask_logit.float()
    if not (t.stride(2) == mw * t.stride(3)):
        t = t.contiguous()
    K = row.shape[0] if K is None else int(K)
    out = torch.empty((K, 1, mh, mw), dtype=torch.float32, device=t.device)
    check(lib().upsnet_mask_logit_gather(stream(), ptr(t), n, C, mh * mw, t.stride(0), t.stride(1), t.stride(3),
                                         ptr(row.to(torch.int32).contiguous()), ptr(cls.to(torch.int64).contiguous()), K, ptr(out)),
          "mask_logit_gather")
    return out


# ----------------------------------------------------------------------------- panoptic head
def mask_removal(mask_rois4, cls_prob, mask_logit, cls_idx, num_thing_classes, im_shape, fraction_threshold=0.3, m_dev=None):
    """Device-side MaskRemoval selection. Returns (keep_inds [m] int64, num_keep, real_keep) device tensors.
    m_dev: optional device int32 count of valid rows (the arrays then have fixed capacity m)."""
    require_cuda(mask_rois4, cls_prob, mask_logit, cls_idx)
    mask_rois4, cls_prob = f32c(mask_rois4), f32c(cls_prob).reshape(-1)
    m = mask_rois4.shape[0]
    mask_logit = f32c(mask_logit).reshape(m, -1)
    ms = int(round(mask_logit.shape[1] ** 0.5))
    cls_idx = cls_idx.to(torch.int64).contiguous().reshape(-1)
    H, W = int(im_shape[0]), int(im_shape[1])
    dev = mask_rois4.device
    keep = torch.empty((m,), dtype=torch.int64, device=dev)   # (the finalize kernel zero-fills the rows past the count)
    num = torch.empty((1,), dtype=torch.int32, device=dev)
    real = torch.empty((1,), dtype=torch.int32, device=dev)
    ws = _ws(lib().upsnet_mask_removal_workspace_bytes(m, num_thing_classes, H, W), dev)
    check(lib().upsnet_mask_removal(stream(), ptr(mask_rois4), ptr(cls_prob), ptr(mask_logit), ptr(cls_idx), m, ptr(m_dev), ms,
                                    int(num_thing_classes), H, W, float(fraction_threshold), ptr(keep), ptr(num), ptr(real),
                                    ptr(ws)), "mask_removal")
    return keep, num, real


def panoptic_tail_pack(keep, num_keep, cls, scores, det_num, pan_num, extra_num):
    """(kept_cls, kept_scores, counters[4] int32) of the fixed-capacity tail in one launch: cls / scores gathered at the kept rows
    (rows past num_keep read row 0) and {det_num, pan_num, extra_num, num_keep} packed for the single host read."""
    require_cuda(keep, cls, scores)
    K = int(cls.shape[0])
    cls, scores = cls.to(torch.int64).contiguous().reshape(-1), f32c(scores).reshape(-1)
    kept_cls = torch.empty((K,), dtype=torch.int64, device=cls.device)
    kept_scores = torch.empty((K,), dtype=torch.float32, device=cls.device)
    counters = torch.empty((4,), dtype=torch.int32, device=cls.device)
    check(lib().upsnet_panoptic_tail_pack(stream(), ptr(keep), ptr(num_keep), K, ptr(cls), ptr(scores), ptr(det_num), ptr(pan_num),
                                          ptr(extra_num), ptr(kept_cls), ptr(kept_scores), ptr(counters)), "panoptic_tail_pack")
    return kept_cls, kept_scores, counters


def mask_paste(mask_rois4, mask_logit, keep, num, real, k, im_shape):
    require_cuda(mask_rois4, mask_logit)
    mask_rois4 = f32c(mask_rois4)
    m = mask_rois4.shape[0]
    mask_logit = f32c(mask_logit).reshape(m, -1)
    ms = int(round(mask_logit.shape[1] ** 0.5))
    H, W = int(im_shape[0]), int(im_shape[1])
    out = torch.empty((1, k, H, W), dtype=torch.float32, device=mask_rois4.device)
    check(lib().upsnet_mask_paste(stream(), ptr(mask_rois4), ptr(mask_logit), ptr(keep), ptr(num), ptr(real), int(k), ms, H, W,
                                  ptr(out)), "mask_paste")
    return out


def seg_term(fcn_output, boxes4, cls, class_map):
    require_cuda(fcn_output, boxes4, cls, class_map)
    fcn = f32c(fcn_output)
    _, S, H, W = fcn.shape
    boxes4 = f32c(boxes4)
    k = boxes4.shape[0]
    out = torch.empty((1, k, H, W), dtype=torch.float32, device=fcn.device)
    check(lib().upsnet_seg_term(stream(), ptr(fcn), S, H, W, ptr(boxes4), ptr(cls.to(torch.int64).contiguous()),
                                ptr(class_map), k, ptr(out)), "seg_term")
    return out


def panoptic_argmax(fcn_output, num_stuff, seg_inst, mask_energy, enable_void=True):
    require_cuda(fcn_output, seg_inst, mask_energy)
    fcn, seg_inst, mask_energy = f32c(fcn_output), f32c(seg_inst), f32c(mask_energy)
    _, S, H, W = fcn.shape
    k = seg_inst.shape[1]
    pan = torch.empty((1, H, W), dtype=torch.int64, device=fcn.device)
    check(lib().upsnet_panoptic_argmax(stream(), ptr(fcn), S, H, W, int(num_stuff), ptr(seg_inst), ptr(mask_energy), k,
                                       int(bool(enable_void)), ptr(pan)), "panoptic_argmax")
    return pan


def panoptic_fuse(fcn_output, num_stuff, mask_rois5, mask_logit, cls_idx, keep, num, real, class_map, enable_void=True,
                  want_sem=True):
    """Fused head: returns (panoptic [1,H,W] int64, semantic argmax [1,H,W] int64 or None)."""
    require_cuda(fcn_output, mask_rois5, mask_logit, cls_idx)
    fcn = f32c(fcn_output)
    _, S, H, W = fcn.shape
    mask_rois5 = f32c(mask_rois5)
    m = mask_rois5.shape[0]
    if m > 256:   # the kernels keep their instance table in LDS (FUSE_MAXK); more rows would be dropped silently
        raise RuntimeError("panoptic_fuse: at most 256 instance rows (got %d); use mask_paste + seg_term + panoptic_argmax" % m)
    mask_logit = f32c(mask_logit).reshape(m, -1)
    ms = int(round(mask_logit.shape[1] ** 0.5))
    pan = torch.empty((1, H, W), dtype=torch.int64, device=fcn.device)
    sem = torch.empty((1, H, W), dtype=torch.int64, device=fcn.device) if want_sem else None
    check(lib().upsnet_panoptic_fuse(stream(), ptr(fcn), S, H, W, int(num_stuff), ptr(mask_rois5), ptr(mask_logit),
                                     ptr(cls_idx.to(torch.int64).contiguous()), ptr(keep), ptr(num), ptr(real), int(min(m, 256)),
                                     ms, ptr(class_map), int(bool(enable_void)), ptr(pan), ptr(sem)), "panoptic_fuse")
    return pan, sem


# ----------------------------------------------------------------------------- dense convolution (fp32 MFMA implicit GEMM)
def conv_supported(cin, kh, kw, groups=1, dilation=(1, 1)):
    return groups == 1 and tuple(dilation) == (1, 1) and cin % 32 == 0 and kh == kw and kh * kw <= 49


def pack_conv_weight(weight):
    """[Cout,Cin,kh,kw] -> ([kh*kw*Cin, ldw], ldw) with ldw = Cout rounded up to 32 (zero padded columns)."""
    require_cuda(weight)
    weight = f32c(weight)
    Cout, Cin, kh, kw = weight.shape
    ldw = (Cout + 31) // 32 * 32
    wp = torch.empty((kh * kw * Cin, ldw), dtype=torch.float32, device=weight.device)
    check(lib().upsnet_conv_pack_weight(stream(), ptr(weight), Cout, Cin, kh, kw, ldw, ptr(wp)), "conv_pack_weight")
    return wp, ldw


def conv2d_nhwc_multi(xs, wpack, ldw, bias, cout, ksize, stride, pad, relu=False, residuals=None, residual_up=False):
    """Up to 5 logical-NCHW tensors through the SAME convolution in one launch; returns channels_last outputs.
    out_i = relu?(conv(x_i) + bias + residual_i); residual_up: residual_i is at half resolution and is added through
    a nearest x2 upsampling (FPN top-down path)."""
    require_cuda(wpack, *xs)
    assert 1 <= len(xs) <= 5
    xs = [nhwc(x.float()) for x in xs]
    cin = xs[0].shape[1]
    outs, ress = [], None
    shapes = []
    for x in xs:
        N, C, H, W = x.shape
        if C != cin:
            raise RuntimeError("conv2d_nhwc_multi: channel mismatch")
        shapes.append((N, (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1, cout))
    # the outputs of a multi-map launch are carved from ONE allocation (outs[0]._ups_flat), so that a following elementwise op over
    # all maps (the RPN's sigmoid over 5 levels) is one launch over the flat buffer instead of one per map
    # (every map starts on a 256-byte boundary of the flat buffer, whatever its channel count: 15-channel RPN heads, 18 / 27-channel
    # offset maps -- a later 16-byte-per-lane consumer must not see a 4-byte-aligned base)
    sizes = [n * h * w * c for n, h, w, c in shapes]
    starts, off = [], 0
    for sz in sizes:
        starts.append(off)
        off += (sz + 63) // 64 * 64
    flat = torch.empty((off,), dtype=torch.float32, device=xs[0].device)
    for (n, h, w, c), sz, st in zip(shapes, sizes, starts):
        outs.append(flat[st:st + sz].view(n, h, w, c).permute(0, 3, 1, 2))
    outs[0]._ups_flat = flat
    if residuals is not None:
        ress = [nhwc(r.float()) for r in residuals]
        for r, o in zip(ress, outs):
            want = (o.shape[0], o.shape[1], o.shape[2] // 2, o.shape[3] // 2) if residual_up else tuple(o.shape)
            if tuple(r.shape) != want or (residual_up and (o.shape[2] % 2 or o.shape[3] % 2)):
                raise RuntimeError("conv2d_nhwc: residual shape %s != %s" % (tuple(r.shape), want))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_nhwc_f32(stream(), len(xs), ptr_array(xs), ptr_array(ress) if ress is not None else None,
                                       ptr_array(outs), int_array([x.shape[0] for x in xs]), int_array([x.shape[2] for x in xs]),
                                       int_array([x.shape[3] for x in xs]), int(cin), ptr(wpack), int(ldw),
                                       ptr(None if bias is None else f32c(bias)), int(cout), int(ksize), int(ksize), int(stride),
                                       int(pad), int(bool(relu)), int(bool(residual_up and ress is not None))), "conv2d_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = sum(o.shape[0] * o.shape[2] * o.shape[3] for o in outs)
        nin = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in xs)
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * ksize * ksize * npix,
                                  4.0 * (cin * nin + cout * npix * (2 if ress is not None else 1) + cout * cin * ksize * ksize),
                                  "direct %dx%d/%d %d->%d %s%s" % (ksize, ksize, stride, cin, cout, [tuple(x.shape[0:1] + x.shape[2:]) for x in xs],
                                                                    " +res" if ress is not None else "")))
    return outs


def conv2d_nhwc_multiw(xs, wpacks, ldw, cout, ksize, stride, pad, relu=False):
    """Up to 5 maps through the same convolution GEOMETRY, each with its own packed weights, in one launch (no bias / residual)."""
    require_cuda(*xs, *wpacks)
    assert 1 <= len(xs) <= 5 and len(wpacks) == len(xs)
    xs = [nhwc(x.float()) for x in xs]
    cin = xs[0].shape[1]
    outs = [_nhwc_out(x.shape[0], cout, (x.shape[2] + 2 * pad - ksize) // stride + 1, (x.shape[3] + 2 * pad - ksize) // stride + 1, x.device)
            for x in xs]
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_nhwc_f32_multiw(stream(), len(xs), ptr_array(xs), ptr_array(outs), int_array([x.shape[0] for x in xs]),
                                              int_array([x.shape[2] for x in xs]), int_array([x.shape[3] for x in xs]), int(cin),
                                              ptr_array(wpacks), int(ldw), int(cout), int(ksize), int(ksize), int(stride), int(pad),
                                              int(bool(relu))), "conv2d_nhwc_f32_multiw")
    if PROFILE['enabled']:
        ev1.record()
        npix = sum(o.shape[0] * o.shape[2] * o.shape[3] for o in outs)
        nin = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in xs)
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * ksize * ksize * npix,
                                  4.0 * (cin * nin + cout * npix + len(xs) * cout * cin * ksize * ksize),
                                  "direct %dx%d/%d %d->%d %s per-map weights" % (ksize, ksize, stride, cin, cout, [tuple(x.shape[2:]) for x in xs])))
    return outs


def conv2d_nhwc(x, wpack, ldw, bias, cout, ksize, stride, pad, relu=False, residual=None, residual_up=False):
    """x: logical NCHW tensor (any batch); returns a channels_last [N,Cout,Ho,Wo] tensor.
    out = relu?(conv(x) + bias + residual) in one kernel."""
    return conv2d_nhwc_multi([x], wpack, ldw, bias, cout, ksize, stride, pad, relu, None if residual is None else [residual],
                             residual_up)[0]


def pack_conv1x1_weight(weight):
    """[Cout,Cin,1,1] -> fragment-order pack for conv1x1_frag (csrc/conv1x1.hip; the layout of csrc/deform_fused.hip with one tap)."""
    require_cuda(weight)
    w = f32c(weight)
    cout, cin = w.shape[0], w.shape[1]
    wp = torch.empty((lib().upsnet_dcn_packed_weight_floats(cout, cin, 1, 1),), dtype=torch.float32, device=w.device)
    check(lib().upsnet_dcn_pack_weight(stream(), ptr(w), cout, cin, 1, 1, ptr(wp)), "dcn_pack_weight")
    return wp


def conv1x1_frag(x, wpack, bias, cout, stride=1, relu=False, residual=None, residual_up=False, out=None, ksplit=1):
    """1x1 convolution (stride 1 / 2) on the lean fp32 MFMA GEMM kernel of csrc/conv1x1.hip: out = relu?(conv(x) + bias + residual),
    residual_up: the residual is at half resolution and is added through a nearest x2 upsampling (FPN top-down path).
    x: logical NCHW (any batch), returns channels_last [N,Cout,Ho,Wo]. out: caller-provided channels_last output (a view of a larger
    tensor); ksplit > 1: the K walk of every tile split over ksplit workgroups + the reduce kernel (no residual_up)."""
    require_cuda(wpack, x)
    x = nhwc(x.float())
    N, cin, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = _nhwc_out(N, cout, Ho, Wo, x.device)
    elif tuple(out.shape) != (N, cout, Ho, Wo) or out.dtype != torch.float32 or not out.permute(0, 2, 3, 1).is_contiguous():
        raise RuntimeError("conv1x1_frag: out must be a channels_last fp32 [%d,%d,%d,%d]" % (N, cout, Ho, Wo))
    res = None
    if residual is not None:
        res = nhwc(residual.float())
        want = (N, cout, out.shape[2] // 2, out.shape[3] // 2) if residual_up else tuple(out.shape)
        if tuple(res.shape) != want or (residual_up and (out.shape[2] % 2 or out.shape[3] % 2)):
            raise RuntimeError("conv1x1_frag: residual shape %s != %s" % (tuple(res.shape), want))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    if ksplit > 1:
        if residual_up:
            raise RuntimeError("conv1x1_frag: split-K has no upsampled-residual epilogue")
        ws = _ws(lib().upsnet_conv1x1_splitk_workspace_bytes(N, Ho, Wo, int(cout), int(ksplit)), x.device)
        check(lib().upsnet_conv1x1_frag_nhwc_f32_splitk(stream(), ptr(x), ptr(res), ptr(out), N, H, W, int(cin), ptr(wpack),
                                                        ptr(None if bias is None else f32c(bias)), int(cout), int(stride), int(bool(relu)),
                                                        int(ksplit), ptr(ws)), "conv1x1_frag_nhwc_f32_splitk")
    else:
        check(lib().upsnet_conv1x1_frag_nhwc_f32(stream(), ptr(x), ptr(res), ptr(out), N, H, W, int(cin), ptr(wpack),
                                                 ptr(None if bias is None else f32c(bias)), int(cout), int(stride), int(bool(relu)),
                                                 int(bool(residual_up and res is not None))), "conv1x1_frag_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = out.shape[0] * out.shape[2] * out.shape[3]
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * npix,
                                  4.0 * (cin * npix + cout * npix * (2 if res is not None else 1) + cout * cin),
                                  "direct 1x1/%d %d->%d [%s]%s%s (gemm)" % (stride, cin, cout, tuple(x.shape[0:1] + x.shape[2:]), " +res" if res is not None else "",
                                                                            " split-K x%d" % ksplit if ksplit > 1 else "")))
    return out


def pack_conv1x1_ksw_weight(weight):
    """[Cout,Cin,1,1] -> 16x16x4 fragment-order pack for conv1x1_ksw (csrc/conv1x1_ksw.hip)."""
    require_cuda(weight)
    w = f32c(weight)
    cout, cin = w.shape[0], w.shape[1]
    wp = torch.empty((lib().upsnet_conv1x1_ksw_packed_weight_floats(cout, cin),), dtype=torch.float32, device=w.device)
    check(lib().upsnet_conv1x1_ksw_pack_weight(stream(), ptr(w), cout, cin, ptr(wp)), "conv1x1_ksw_pack_weight")
    return wp


def conv1x1_ksw(x, wpack, bias, cout, tile, stride=1, relu=False, residual=None, out=None, split_n=False):
    """1x1 convolution (stride 1 / 2) on the small-tile kernel of csrc/conv1x1_ksw.hip (16x16x4 MFMA fragments, K split over the waves):
    out = relu?(conv(x) + bias + residual). tile = (pixels, channels) of a workgroup: {(16, 64), (32, 32), (32, 64), (64, 64)} with the waves
    splitting K, {(16, 256), (32, 128), (32, 256)} with split_n (the waves split the channels, each walks all of K).
    x: logical NCHW (any batch), returns channels_last [N,Cout,Ho,Wo]."""
    require_cuda(wpack, x)
    x = nhwc(x.float())
    N, cin, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = _nhwc_out(N, cout, Ho, Wo, x.device)
    elif tuple(out.shape) != (N, cout, Ho, Wo) or out.dtype != torch.float32 or not out.permute(0, 2, 3, 1).is_contiguous():
        raise RuntimeError("conv1x1_ksw: out must be a channels_last fp32 [%d,%d,%d,%d]" % (N, cout, Ho, Wo))
    res = None
    if residual is not None:
        res = nhwc(residual.float())
        if tuple(res.shape) != tuple(out.shape):
            raise RuntimeError("conv1x1_ksw: residual shape %s != %s" % (tuple(res.shape), tuple(out.shape)))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv1x1_ksw_nhwc_f32(stream(), ptr(x), ptr(res), ptr(out), N, H, W, int(cin), ptr(wpack),
                                            ptr(None if bias is None else f32c(bias)), int(cout), int(stride), int(bool(relu)),
                                            int(tile[0]), int(tile[1]), int(bool(split_n))), "conv1x1_ksw_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = out.shape[0] * out.shape[2] * out.shape[3]
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * npix,
                                  4.0 * (cin * npix + cout * npix * (2 if res is not None else 1) + cout * cin),
                                  "direct 1x1/%d %d->%d [%s]%s (gemm ksw %dx%d%s)" % (stride, cin, cout, tuple(x.shape[0:1] + x.shape[2:]),
                                                                                     " +res" if res is not None else "", tile[0], tile[1], "n" if split_n else "k")))
    return out


def pack_conv3x3_ksw_weight(weight):
    """[Cout <= 32, Cin, 3, 3] -> 16x16x4 fragment-order pack for conv3x3_ksw (csrc/conv1x1_ksw.hip)."""
    require_cuda(weight)
    w = f32c(weight)
    cout, cin = w.shape[0], w.shape[1]
    wp = torch.empty((lib().upsnet_conv3x3_ksw_packed_weight_floats(cin),), dtype=torch.float32, device=w.device)
    check(lib().upsnet_conv3x3_ksw_pack_weight(stream(), ptr(w), cout, cin, ptr(wp)), "conv3x3_ksw_pack_weight")
    return wp


def conv3x3_ksw(x, wpack, bias, cout, relu=False):
    """3x3 / stride 1 / pad 1 convolution into cout <= 32 channels on the small-tile kernel (csrc/conv1x1_ksw.hip, conv3x3_ksw_f32_kernel):
    the offset predictors of the deformable bottlenecks on small maps. x: logical NCHW, returns channels_last [N,cout,H,W]."""
    require_cuda(wpack, x)
    x = nhwc(x.float())
    N, cin, H, W = x.shape
    out = _nhwc_out(N, cout, H, W, x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv3x3_ksw_nhwc_f32(stream(), ptr(x), ptr(out), N, H, W, int(cin), ptr(wpack), ptr(None if bias is None else f32c(bias)),
                                            int(cout), int(bool(relu))), "conv3x3_ksw_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * H * W
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * 9 * npix, 4.0 * (cin * npix + cout * npix + cout * cin * 9),
                                  "direct 3x3/1 %d->%d [%s] (ksw 16x32)" % (cin, cout, (N, H, W))))
    return out


def conv1x1_siblings(x, wpack, bias, cout_a, cout_b, stride=1, relu_a=True, relu_b=False):
    """Two 1x1 convolutions of the same input in one launch (csrc/conv1x1.hip, sibling mode): rows [0, cout_a) of the concatenated weight
    -> out_a (ReLU flag relu_a), the remaining cout_b rows -> out_b. wpack: pack_conv1x1_weight(torch.cat([w_a, w_b])); bias: both biases
    concatenated or None. Bit-identical to two conv1x1_frag calls. Returns two channels_last tensors."""
    require_cuda(wpack, x)
    x = nhwc(x.float())
    N, cin, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out_a, out_b = _nhwc_out(N, cout_a, Ho, Wo, x.device), _nhwc_out(N, cout_b, Ho, Wo, x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv1x1_siblings_nhwc_f32(stream(), ptr(x), ptr(out_a), ptr(out_b), N, H, W, int(cin), ptr(wpack),
                                                 ptr(None if bias is None else f32c(bias)), int(cout_a), int(cout_b), int(stride),
                                                 int(bool(relu_a)), int(bool(relu_b))), "conv1x1_siblings_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix, cout = N * Ho * Wo, cout_a + cout_b
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * npix, 4.0 * (cin * npix + cout * npix + cout * cin),
                                  "direct 1x1/%d siblings %d->%d+%d [%s] (gemm)" % (stride, cin, cout_a, cout_b, (N, H, W))))
    return out_a, out_b


PAIR32_WAVES = int(os.environ.get('UPSNET_CONV1X1_PAIR32_WAVES', '8'))   # waves per workgroup of the 32-pixel pair kernel (res4): 8 (two per SIMD) or 4
_pair32_waves_set = [None]


def conv1x1_pair(x, residual, w3pack, bias3, c1, w1pack, bias1, c2):
    """Tail of one bottleneck and head of the next in one launch (csrc/conv1x1_pair.hip):
    out1 = relu(conv1x1(x; w3) + bias3 + residual), out2 = relu(conv1x1(out1; w1) + bias1); both bit-identical to two
    conv1x1_frag calls. x: logical NCHW; returns two channels_last tensors."""
    require_cuda(w3pack, w1pack, x, residual)
    x, res = nhwc(x.float()), nhwc(residual.float())
    N, c0, H, W = x.shape
    if tuple(res.shape) != (N, c1, H, W):
        raise RuntimeError("conv1x1_pair: residual shape %s != %s" % (tuple(res.shape), (N, c1, H, W)))
    out1, out2 = _nhwc_out(N, c1, H, W, x.device), _nhwc_out(N, c2, H, W, x.device)
    if _pair32_waves_set[0] != PAIR32_WAVES:
        lib().upsnet_conv1x1_pair32_tuning(PAIR32_WAVES)
        _pair32_waves_set[0] = PAIR32_WAVES
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv1x1_pair_nhwc_f32(stream(), ptr(x), ptr(res), ptr(out1), ptr(out2), N * H * W, int(c0), ptr(w3pack),
                                             ptr(None if bias3 is None else f32c(bias3)), int(c1), ptr(w1pack),
                                             ptr(None if bias1 is None else f32c(bias1)), int(c2)), "conv1x1_pair_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * H * W
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * (c1 * c0 + c2 * c1) * npix,
                                  4.0 * (c0 * npix + 2 * c1 * npix + c2 * npix + c1 * c0 + c2 * c1),
                                  "direct 1x1 pair %d->%d->%d [%s] +res (gemm)" % (c0, c1, c2, (N, H, W))))
    return out1, out2


def fcn_score_combine(parts, bias=None):
    """score = bias + parts[0] + sum_l bilinear_up_{2^l}(parts[l]); parts[l]: logical NCHW [1,S,H>>l,W>>l] (channels_last
    memory). Returns a channels_last [1,S,H,W] tensor (fcn.py:94-100 with the 1x1 conv commuted below the upsampling)."""
    require_cuda(*parts)
    parts = [nhwc(t.float()) for t in parts]
    _, S, H, W = parts[0].shape
    for l, t in enumerate(parts):
        if tuple(t.shape) != (1, S, H >> l, W >> l):
            raise RuntimeError("fcn_score_combine: level %d has shape %s, expected %s" % (l, tuple(t.shape), (1, S, H >> l, W >> l)))
    out = _nhwc_out(1, S, H, W, parts[0].device)
    check(lib().upsnet_fcn_score_combine(stream(), len(parts), ptr_array(parts), S, H, W, ptr(None if bias is None else f32c(bias)),
                                         ptr(out)), "fcn_score_combine")
    return out


def panoptic_fuse_up(fcn_score, scale, num_stuff, mask_rois5, mask_logit, cls_idx, keep, num, real, class_map, want_sem=True):
    """Fused head INCLUDING FCNHead's x4 bilinear upsampling: takes the low-resolution fcn_score [1,S,Hs,Ws]
    (logical NCHW; channels_last memory is consumed as is) and returns (panoptic, semantic) [1,Hs*scale,Ws*scale] int64."""
    require_cuda(fcn_score, mask_rois5, mask_logit, cls_idx)
    sc = fcn_score.float()
    _, S, Hs, Ws = sc.shape
    is_nhwc = (not sc.is_contiguous()) and sc.is_contiguous(memory_format=torch.channels_last)
    sc = sc if is_nhwc else sc.contiguous()
    mask_rois5 = f32c(mask_rois5)
    m = mask_rois5.shape[0]
    if m > 256:   # the kernels keep their instance table in LDS (FUSE_MAXK); more rows would be dropped silently
        raise RuntimeError("panoptic_fuse: at most 256 instance rows (got %d); use mask_paste + seg_term + panoptic_argmax" % m)
    mask_logit = f32c(mask_logit).reshape(m, -1)
    ms = int(round(mask_logit.shape[1] ** 0.5))
    H, W = Hs * scale, Ws * scale
    pan = torch.empty((1, H, W), dtype=torch.int64, device=sc.device)
    sem = torch.empty((1, H, W), dtype=torch.int64, device=sc.device) if want_sem else None
    check(lib().upsnet_panoptic_fuse_up(stream(), ptr(sc), int(is_nhwc), S, Hs, Ws, int(scale), int(num_stuff), ptr(mask_rois5),
                                        ptr(mask_logit), ptr(cls_idx.to(torch.int64).contiguous()), ptr(keep), ptr(num), ptr(real),
                                        int(min(m, 256)), ms, ptr(class_map), ptr(pan), ptr(sem)), "panoptic_fuse_up")
    return pan, sem


# ----------------------------------------------------------------------------- stem / deconvolution / input blob
def pack_stem_weight(weight):
    """[Cout,Cin<=4,KH,KW<=8] -> ([KH*32, ldw], ldw): one K slab = one kernel row of 8 pixels x 4 channels."""
    require_cuda(weight)
    weight = f32c(weight)
    Cout, Cin, kh, kw = weight.shape
    ldw = (Cout + 31) // 32 * 32
    wp = torch.empty((kh * 32, ldw), dtype=torch.float32, device=weight.device)
    check(lib().upsnet_conv_pack_weight_stem(stream(), ptr(weight), Cout, Cin, kh, kw, ldw, ptr(wp)), "conv_pack_weight_stem")
    return wp, ldw


def pack_stem_pool_weight_f32(weight):
    """[64, Cin <= 3, 7, 7] fp32 -> the operand pack of stem_pool_f32 (2 x 75 x 64 floats)."""
    require_cuda(weight)
    weight = f32c(weight)
    if tuple(weight.shape[0:1] + weight.shape[2:]) != (64, 7, 7) or weight.shape[1] > 3:
        raise RuntimeError("pack_stem_pool_weight_f32: weight must be [64, Cin <= 3, 7, 7]")
    wp = torch.empty(2 * 75 * 64, dtype=torch.float32, device=weight.device)
    check(lib().upsnet_stem_pool_pack_weight_f32(stream(), ptr(weight), int(weight.shape[1]), ptr(wp)), "stem_pool_pack_weight_f32")
    return wp


def stem_pool_f32(x4, wpack, bias):
    """max_pool2d(relu(conv7x7/2/3(x) + bias), 3, 2, 1) in one launch; x4: logical [N,4,H,W] channels_last fp32 (image_to_nhwc4 /
    prep_image_u8). Returns channels_last fp32 [N,64,Hp,Wp]."""
    require_cuda(x4, wpack)
    x4 = nhwc(x4.float())
    N, C, H, W = x4.shape
    if C != 4:
        raise RuntimeError("stem_pool_f32: input must have 4 channels (RGB + zero), got %d" % C)
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = _nhwc_out(N, 64, Hp, Wp, x4.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_stem_pool_f32(stream(), ptr(x4), N, H, W, ptr(wpack), ptr(None if bias is None else f32c(bias)), ptr(out)), "stem_pool_f32")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * 64 * 147 * N * Hc * Wc, 16.0 * N * H * W + 4.0 * 64 * N * Hp * Wp + 4.0 * 64 * 147, 'stem + pool'))
    return out


def pack_stem_pool_weight_bf16(weight):
    """[64, Cin <= 4, 7, 7] fp32 -> the bf16 fragment pack of stem_pool_bf16 (2 x 14 x 64 x 8 elements)."""
    require_cuda(weight)
    weight = f32c(weight)
    if tuple(weight.shape[0:1] + weight.shape[2:]) != (64, 7, 7) or weight.shape[1] > 4:
        raise RuntimeError("pack_stem_pool_weight_bf16: weight must be [64, Cin <= 4, 7, 7]")
    wp = torch.empty(2 * 14 * 64 * 8, dtype=torch.bfloat16, device=weight.device)
    check(lib().upsnet_stem_pool_pack_weight_bf16(stream(), ptr(weight), int(weight.shape[1]), ptr(wp)), "stem_pool_pack_weight_bf16")
    return wp


def stem_pool_bf16(x4, wpack, bias):
    """relu(conv7x7/2/3(x) + bias) -> max_pool 3x3/2/1 in one launch; x4: logical [N,4,H,W] channels_last fp32 (image_to_nhwc4 /
    prep_image_u8). Returns a bf16 channels_last [N,64,Hp,Wp]."""
    require_cuda(x4, wpack)
    x4 = nhwc(x4.float())
    N, C, H, W = x4.shape
    if C != 4:
        raise RuntimeError("stem_pool_bf16: input must have 4 channels (RGB + zero), got %d" % C)
    Hc, Wc = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    Hp, Wp = (Hc - 1) // 2 + 1, (Wc - 1) // 2 + 1
    out = torch.empty((N, Hp, Wp, 64), dtype=torch.bfloat16, device=x4.device).permute(0, 3, 1, 2)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_stem_pool_bf16(stream(), ptr(x4), N, H, W, ptr(wpack), ptr(None if bias is None else f32c(bias)), ptr(out)), "stem_pool_bf16")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv_bf16', ev0, ev1, 2.0 * 64 * 147 * N * Hc * Wc, 16.0 * N * H * W + 2.0 * 64 * N * Hp * Wp))
    return out


def image_to_nhwc4(x):
    """fp32 [N,C<=4,H,W] -> physical [N,H,W,4] (returned as a logical [N,4,H,W] channels_last tensor)."""
    require_cuda(x)
    x = f32c(x)
    N, C, H, W = x.shape
    out = _nhwc_out(N, 4, H, W, x.device)
    check(lib().upsnet_image_to_nhwc4(stream(), ptr(x), N, C, H, W, ptr(out)), "image_to_nhwc4")
    return out


def conv2d_stem(x4, wpack, ldw, bias, cout, kh, kw, stride, pad, relu=False):
    """x4: logical [N,4,H,W] channels_last (from image_to_nhwc4 / prep_image_u8). Returns channels_last [N,Cout,Ho,Wo]."""
    require_cuda(x4, wpack)
    x4 = nhwc(x4.float())
    N, C, H, W = x4.shape
    if C != 4:
        raise RuntimeError("conv2d_stem: input must have 4 channels (RGB + zero), got %d" % C)
    Ho, Wo = (H + 2 * pad - kh) // stride + 1, (W + 2 * pad - kw) // stride + 1
    out = _nhwc_out(N, cout, Ho, Wo, x4.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_stem_nhwc4_f32(stream(), ptr(x4), N, H, W, ptr(wpack), int(ldw), ptr(None if bias is None else f32c(bias)),
                                             int(cout), int(kh), int(kw), int(stride), int(pad), int(bool(relu)), ptr(out)),
          "conv2d_stem_nhwc4_f32")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * 3 * kh * kw * N * Ho * Wo, 4.0 * (4 * N * H * W + cout * N * Ho * Wo + cout * 3 * kh * kw)))
    return out


def pack_deconv2x2_weight(weight):
    """ConvTranspose2d weight [Cin,Cout,2,2] -> ([Cin, ldw], ldw) with columns (dy,dx,co)."""
    require_cuda(weight)
    weight = f32c(weight)
    Cin, Cout, kh, kw = weight.shape
    if (kh, kw) != (2, 2):
        raise RuntimeError("pack_deconv2x2_weight: kernel must be 2x2")
    ldw = (4 * Cout + 31) // 32 * 32
    wp = torch.empty((Cin, ldw), dtype=torch.float32, device=weight.device)
    check(lib().upsnet_deconv2x2_pack_weight(stream(), ptr(weight), Cin, Cout, ldw, ptr(wp)), "deconv2x2_pack_weight")
    return wp, ldw


def deconv2x2(x, wpack, ldw, bias, cout, relu=False):
    """ConvTranspose2d(k=2, s=2, p=0) (+bias, +ReLU) on a logical NCHW tensor; returns channels_last [N,Cout,2H,2W]."""
    require_cuda(x, wpack)
    x = nhwc(x.float())
    N, C, H, W = x.shape
    out = _nhwc_out(N, cout, 2 * H, 2 * W, x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_deconv2x2_nhwc_f32(stream(), ptr(x), N, H, W, C, ptr(wpack), int(ldw), ptr(None if bias is None else f32c(bias)),
                                          int(cout), int(bool(relu)), ptr(out)), "deconv2x2_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * 4 * cout * C * N * H * W, 4.0 * (C * N * H * W + 4 * cout * N * H * W + 4 * cout * C)))
    return out


def pack_deconv2x2_weight_frag(weight, bias):
    """ConvTranspose2d weight [Cin,Cout,2,2] (+ bias [Cout]) -> (fragment-order pack of the [4 Cout, Cin] matrix with rows (dy, dx, co),
    bias repeated per (dy, dx)) for deconv2x2_frag (csrc/conv1x1.hip, scatter epilogue)."""
    require_cuda(weight)
    Cin, Cout, kh, kw = weight.shape
    if (kh, kw) != (2, 2) or Cin % 32 or Cout % 32:
        raise RuntimeError("pack_deconv2x2_weight_frag: kernel 2x2, Cin % 32 == 0, Cout % 32 == 0")
    w = weight.detach().float().permute(2, 3, 1, 0).reshape(4 * Cout, Cin, 1, 1).contiguous()
    return pack_conv1x1_weight(w), (None if bias is None else bias.detach().float().repeat(4).contiguous())


def deconv2x2_frag(x, wpack, bias4, cout, relu=False):
    """ConvTranspose2d(k=2, s=2, p=0) (+bias, +ReLU) on the lean GEMM kernel; returns channels_last [N,Cout,2H,2W]."""
    require_cuda(x, wpack)
    x = nhwc(x.float())
    N, C, H, W = x.shape
    out = _nhwc_out(N, cout, 2 * H, 2 * W, x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_deconv2x2_frag_nhwc_f32(stream(), ptr(x), ptr(out), N, H, W, C, ptr(wpack), ptr(bias4), int(cout), int(bool(relu))),
          "deconv2x2_frag_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * 4 * cout * C * N * H * W, 4.0 * (C * N * H * W + 4 * cout * N * H * W + 4 * cout * C),
                                  "deconv 2x2/2 %d->%d [%s] (gemm)" % (C, cout, (N, H, W))))
    return out


def pack_deconv2x2_weight_bf16(weight):
    """ConvTranspose2d weight [Cin,Cout,2,2] -> (bf16 pack, ldw) of the [4 Cout, Cin] matrix with rows (ky, kx, co) for deconv2x2_bf16."""
    require_cuda(weight)
    Cin, Cout, kh, kw = weight.shape
    if (kh, kw) != (2, 2) or Cin % 64 or Cout % 32:
        raise RuntimeError("pack_deconv2x2_weight_bf16: kernel 2x2, Cin % 64 == 0, Cout % 32 == 0")
    w = weight.detach().float().permute(2, 3, 1, 0).reshape(4 * Cout, Cin, 1, 1).contiguous()
    hi, _, ldw = pack_conv_weight_bf16(w, split=False)
    return hi, ldw


def deconv2x2_bf16(x, whi, ldw, bias, cout, relu=False, out_dtype=torch.float32):
    """ConvTranspose2d(k=2, s=2, p=0) (+bias, +ReLU) of a bf16 map on the bf16 matrix cores; channels_last [N,Cout,2H,2W]."""
    require_cuda(x, whi)
    if x.dtype != torch.bfloat16:
        raise RuntimeError("deconv2x2_bf16: expects a bf16 map")
    x = nhwc(x)
    N, C, H, W = x.shape
    out16 = out_dtype == torch.bfloat16
    out = (torch.empty((N, 2 * H, 2 * W, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2) if out16
           else _nhwc_out(N, cout, 2 * H, 2 * W, x.device))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_deconv2x2_nhwc_bf16(stream(), ptr(x), N, H, W, C, ptr(whi), int(ldw), ptr(None if bias is None else f32c(bias)),
                                           int(cout), int(bool(relu)), ptr(out), int(out16)), "deconv2x2_nhwc_bf16")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('conv_bf16', ev0, ev1, 2.0 * 4 * cout * C * N * H * W,
                                  2.0 * C * N * H * W + (2.0 if out16 else 4.0) * 4 * cout * N * H * W + 2.0 * 4 * cout * C))
    return out


def prep_image_u8(image_hwc, pixel_means, im_scale, resized_hw, padded_hw, nhwc4=True):
    """uint8 [H,W,3] device image -> padded fp32 blob: logical [1,4,Hp,Wp] channels_last (nhwc4) or [1,3,Hp,Wp] NCHW."""
    import ctypes
    require_cuda(image_hwc)
    if image_hwc.dtype != torch.uint8 or image_hwc.dim() != 3 or image_hwc.shape[2] != 3:
        raise RuntimeError("prep_image_u8: expected a uint8 [H,W,3] image")
    im = image_hwc.contiguous()
    H, W = im.shape[:2]
    (Hr, Wr), (Hp, Wp) = resized_hw, padded_hw
    means = (ctypes.c_double * 3)(*[float(m) for m in pixel_means])
    out = _nhwc_out(1, 4, Hp, Wp, im.device) if nhwc4 else torch.empty((1, 3, Hp, Wp), dtype=torch.float32, device=im.device)
    check(lib().upsnet_prep_image_u8(stream(), ptr(im), H, W, ctypes.cast(means, ctypes.c_void_p), float(im_scale), int(Hr), int(Wr),
                                     int(Hp), int(Wp), int(bool(nhwc4)), ptr(out)), "prep_image_u8")
    return out


def unified_pan_result(pan, seg, cls_inds, id_last_stuff, num_seg_classes, stuff_area_limit=4 * 64 * 64):
    """get_unified_pan_result of ONE image on the device: pan / seg int64 [H,W] (or [1,H,W]), cls_inds int64 [k] -> uint8 [H,W,3]."""
    require_cuda(pan, seg)
    pan = pan.reshape(pan.shape[-2], pan.shape[-1]).to(torch.int64).contiguous()
    seg = seg.reshape(seg.shape[-2], seg.shape[-1]).to(torch.int64).contiguous()
    if pan.shape != seg.shape:
        raise RuntimeError("unified_pan_result: pan %s and seg %s differ in shape" % (tuple(pan.shape), tuple(seg.shape)))
    cls_inds = cls_inds.to(device=pan.device, dtype=torch.int64).contiguous().reshape(-1)
    H, W = pan.shape
    ws = torch.empty((int(lib().upsnet_unified_pan_workspace_bytes()),), dtype=torch.uint8, device=pan.device)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device=pan.device)
    check(lib().upsnet_unified_pan_result(stream(), ptr(pan), ptr(seg), ptr(cls_inds) if cls_inds.numel() else None, int(cls_inds.numel()),
                                          H, W, int(id_last_stuff), int(num_seg_classes), int(stuff_area_limit), ptr(ws), ptr(out)),
          "unified_pan_result")
    return out


def mask_roi_dedup(a_src, a_cls, a_num, b_src, b_cls, b_boxes, b_num):
    """Rows of the B detections inside [A ; unmatched of B] (matching on (source ROI, class)). Returns (map int32 [capB],
    extra_boxes [capB,5], num_extra device int32 [1])."""
    require_cuda(a_src, b_src, b_boxes)
    cap_a, cap_b = a_src.shape[0], b_src.shape[0]
    dev = b_boxes.device
    mp = torch.empty((cap_b,), dtype=torch.int32, device=dev)
    extra = torch.empty((cap_b, 5), dtype=torch.float32, device=dev)
    n_extra = torch.empty((1,), dtype=torch.int32, device=dev)
    check(lib().upsnet_mask_roi_dedup(stream(), ptr(a_src), ptr(a_cls), ptr(a_num), cap_a, ptr(b_src), ptr(b_cls), ptr(f32c(b_boxes)),
                                      ptr(b_num), cap_b, ptr(mp), ptr(extra), ptr(n_extra)), "mask_roi_dedup")
    return mp, extra, n_extra


# ----------------------------------------------------------------------------- Winograd F(2x2,3x3)
def pack_winograd_weight(weight, tn32=False):
    """[Cout,Cin,3,3] -> (U = G g G^T in the kernel's fragment order, 16*Cin*ldw floats, ldw). tn32: the order of the 32-channel
    workgroup form for any Cout (conv2d_winograd_multi(..., tn32=True))."""
    require_cuda(weight)
    weight = f32c(weight)
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3):
        raise RuntimeError("pack_winograd_weight: kernel must be 3x3")
    if tn32:
        ldw = (Cout + 31) // 32 * 32
        wp = torch.empty((16 * Cin, ldw), dtype=torch.float32, device=weight.device)
        check(lib().upsnet_conv_pack_weight_winograd_tn32(stream(), ptr(weight), Cout, Cin, ldw, ptr(wp)), "conv_pack_weight_winograd_tn32")
        return wp, ldw
    ldw = 32 if Cout <= 32 else (Cout + 63) // 64 * 64   # 64 output channels per workgroup; narrow heads: the 32-channel form
    wp = torch.empty((16 * Cin, ldw), dtype=torch.float32, device=weight.device)
    check(lib().upsnet_conv_pack_weight_winograd(stream(), ptr(weight), Cout, Cin, ldw, ptr(wp)), "conv_pack_weight_winograd")
    return wp, ldw


def conv2d_winograd_splitk(x, wpack, ldw, bias, cout, ksplit, relu=False, residual=None):
    """conv2d_winograd for one small map with the K walk split over `ksplit` workgroups per tile (+ the shared reduce kernel)."""
    require_cuda(x, wpack)
    x = nhwc(x.float())
    N, C, H, W = x.shape
    out = _nhwc_out(N, cout, H, W, x.device)
    res = None
    if residual is not None:
        res = nhwc(residual.float())
        if tuple(res.shape) != tuple(out.shape):
            raise RuntimeError("conv2d_winograd_splitk: residual shape %s != %s" % (tuple(res.shape), tuple(out.shape)))
    ws = _ws(lib().upsnet_conv2d_splitk_workspace_bytes(N, H, W, int(cout), 3, 3, 1, 1, int(ksplit)), x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_winograd_nhwc_f32_splitk(stream(), ptr(x), ptr(res), ptr(out), N, H, W, C, ptr(wpack), int(ldw),
                                                       ptr(None if bias is None else f32c(bias)), int(cout), int(bool(relu)), int(ksplit),
                                                       ptr(ws)), "conv2d_winograd_nhwc_f32_splitk")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * H * W
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * C * 9 * npix,
                                  4.0 * (C * npix + cout * npix * (2 if res is not None else 1) + cout * C * 9),
                                  "winograd split-K x%d 3x3/1 %d->%d %s" % (ksplit, C, cout, (N, H, W))))
    return out


def conv2d_winograd_tail(x, wpack, ldw, wpack32, ldw32, bias, cout, n_main, relu=False):
    """3x3 / stride 1 / pad 1 Winograd convolution of a batch in ONE launch of two workgroup forms (csrc/conv_wino.hip,
    conv_wino16_tail_f32_kernel): images [0, n_main) on 32 x 64 tiles, the rest on 32 x 32 tiles (wpack32 from
    pack_winograd_weight(w, tn32=True)). Bit-identical to conv2d_winograd_multi([x], ...)."""
    require_cuda(wpack, wpack32, x)
    x = nhwc(x.float())
    N, C, H, W = x.shape
    out = _nhwc_out(N, cout, H, W, x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_winograd_nhwc_f32_tail(stream(), ptr(x), ptr(out), int(N), int(n_main), int(H), int(W), int(C), ptr(wpack), int(ldw),
                                                     ptr(wpack32), int(ldw32), ptr(None if bias is None else f32c(bias)), int(cout),
                                                     int(bool(relu))), "conv2d_winograd_nhwc_f32_tail")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * H * W
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * C * 9 * npix, 4.0 * (C * npix + cout * npix + cout * C * 9),
                                  "winograd 3x3/1 %d->%d [%s] tail %d" % (C, cout, (N, H, W), N - n_main)))
    return out


def conv2d_winograd_multi(xs, wpack, ldw, bias, cout, relu=False, residuals=None, outs=None, tn32=False):
    """3x3 / stride 1 / pad 1 convolution of up to 5 maps (shared weights) by fused Winograd F(2x2,3x3); same contract as
    conv2d_nhwc_multi. outs: caller-provided channels_last outputs (e.g. batch slices of one tensor); tn32: the 32-channel workgroup
    form (wpack from pack_winograd_weight(w, tn32=True))."""
    require_cuda(wpack, *xs)
    assert 1 <= len(xs) <= 5
    xs = [nhwc(x.float()) for x in xs]
    cin = xs[0].shape[1]
    given, outs, ress = outs, [], None
    for i, x in enumerate(xs):
        N, C, H, W = x.shape
        if C != cin:
            raise RuntimeError("conv2d_winograd_multi: channel mismatch")
        if given is not None:
            o = given[i]
            if tuple(o.shape) != (N, cout, H, W) or o.dtype != torch.float32 or not o.permute(0, 2, 3, 1).is_contiguous():
                raise RuntimeError("conv2d_winograd_multi: outs[%d] must be a channels_last fp32 [%d,%d,%d,%d]" % (i, N, cout, H, W))
            outs.append(o)
        else:
            outs.append(_nhwc_out(N, cout, H, W, x.device))
    if residuals is not None:
        ress = [nhwc(r.float()) for r in residuals]
        for r, o in zip(ress, outs):
            if tuple(r.shape) != tuple(o.shape):
                raise RuntimeError("conv2d_winograd: residual shape %s != %s" % (tuple(r.shape), tuple(o.shape)))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    fn = lib().upsnet_conv2d_winograd_nhwc_f32_tn32 if tn32 else lib().upsnet_conv2d_winograd_nhwc_f32
    check(fn(stream(), len(xs), ptr_array(xs), ptr_array(ress) if ress is not None else None,
             ptr_array(outs), int_array([x.shape[0] for x in xs]), int_array([x.shape[2] for x in xs]),
             int_array([x.shape[3] for x in xs]), int(cin), ptr(wpack), int(ldw),
             ptr(None if bias is None else f32c(bias)), int(cout), int(bool(relu))),
          "conv2d_winograd_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = sum(o.shape[0] * o.shape[2] * o.shape[3] for o in outs)
        # algorithmic work of the convolution (direct-form flops), as for the direct kernel
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * 9 * npix,
                                  4.0 * (cin * npix + cout * npix * (2 if ress is not None else 1) + cout * cin * 9),
                                  "winograd 3x3/1 %d->%d %s%s%s" % (cin, cout, [tuple(x.shape[0:1] + x.shape[2:]) for x in xs],
                                                                    " +res" if ress is not None else "", " tn32" if tn32 else "")))
    return outs


def pack_winograd36_weight(weight):
    """[Cout,Cin,3,3] -> (U = G g G^T of Winograd F(4x4,3x3) in the fragment order of csrc/conv_wino36.hip, 36*Cin*ldw floats, ldw)."""
    require_cuda(weight)
    weight = f32c(weight)
    Cout, Cin, kh, kw = weight.shape
    if (kh, kw) != (3, 3) or Cin % 32:
        raise RuntimeError("pack_winograd36_weight: 3x3 kernel and Cin %% 32 == 0")
    ldw = (Cout + 63) // 64 * 64
    wp = torch.empty((36 * Cin, ldw), dtype=torch.float32, device=weight.device)
    check(lib().upsnet_conv_pack_weight_winograd36(stream(), ptr(weight), Cout, Cin, ldw, ptr(wp)), "conv_pack_weight_winograd36")
    return wp, ldw


def conv2d_winograd36_multi(xs, wpack, ldw, bias, cout, relu=False, outs=None):
    """3x3 / stride 1 / pad 1 convolution (+ bias, ReLU) of up to 5 maps (shared weights) by Winograd F(4x4,3x3): the contract of
    conv2d_winograd_multi without residuals (csrc/conv_wino36.hip)."""
    require_cuda(wpack, *xs)
    assert 1 <= len(xs) <= 5
    xs = [nhwc(x.float()) for x in xs]
    cin = xs[0].shape[1]
    given, outs = outs, []
    for i, x in enumerate(xs):
        N, C, H, W = x.shape
        if C != cin:
            raise RuntimeError("conv2d_winograd36_multi: channel mismatch")
        if given is not None:
            o = given[i]
            if tuple(o.shape) != (N, cout, H, W) or o.dtype != torch.float32 or not o.permute(0, 2, 3, 1).is_contiguous():
                raise RuntimeError("conv2d_winograd36_multi: outs[%d] must be a channels_last fp32 [%d,%d,%d,%d]" % (i, N, cout, H, W))
            outs.append(o)
        else:
            outs.append(_nhwc_out(N, cout, H, W, x.device))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_winograd36_nhwc_f32(stream(), len(xs), ptr_array(xs), ptr_array(outs), int_array([x.shape[0] for x in xs]),
                                                  int_array([x.shape[2] for x in xs]), int_array([x.shape[3] for x in xs]), int(cin), ptr(wpack),
                                                  int(ldw), ptr(None if bias is None else f32c(bias)), int(cout), int(bool(relu))),
          "conv2d_winograd36_nhwc_f32")
    if PROFILE['enabled']:
        ev1.record()
        npix = sum(o.shape[0] * o.shape[2] * o.shape[3] for o in outs)
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * 9 * npix, 4.0 * (cin * npix + cout * npix + cout * cin * 9),
                                  "winograd36 3x3/1 %d->%d %s" % (cin, cout, [tuple(x.shape[0:1] + x.shape[2:]) for x in xs])))
    return outs


def conv2d_winograd36_splitk(x, wpack, ldw, bias, cout, ksplit, relu=False):
    """conv2d_winograd36_multi for ONE map with the channel walk of every tile split over `ksplit` workgroups + the reduce / epilogue kernel
    (csrc/conv_wino36.hip, r13): maps with fewer 32-tile x 64-channel workgroups than CUs."""
    require_cuda(wpack, x)
    x = nhwc(x.float())
    N, cin, H, W = x.shape
    out = _nhwc_out(N, cout, H, W, x.device)
    ws = _ws(lib().upsnet_conv2d_winograd36_splitk_workspace_bytes(N, H, W, int(cout), int(ksplit)), x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_winograd36_nhwc_f32_splitk(stream(), ptr(x), ptr(out), N, H, W, int(cin), ptr(wpack), int(ldw),
                                                         ptr(None if bias is None else f32c(bias)), int(cout), int(bool(relu)), int(ksplit), ptr(ws)),
          "conv2d_winograd36_nhwc_f32_splitk")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * H * W
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * cin * 9 * npix, 4.0 * (cin * npix + cout * npix + cout * cin * 9),
                                  "winograd36 split-K x%d 3x3/1 %d->%d %s" % (ksplit, cin, cout, (N, H, W))))
    return out


# ----------------------------------------------------------------------------- bf16 matrix-core convolution (opt-in)
def pack_conv_weight_bf16(weight, split=True):
    """[Cout,Cin,kh,kw] fp32 -> (hi, lo, ldw): bf16 [kh*kw*Cin/32, ldw, 32] (as int16 storage), ldw = Cout rounded up to 64.
    split=False ("bf16" mode) leaves lo = None."""
    require_cuda(weight)
    weight = f32c(weight)
    Cout, Cin, kh, kw = weight.shape
    ldw = (Cout + 127) // 128 * 128
    shape = (kh * kw * Cin // 32, ldw, 32)
    hi = torch.empty(shape, dtype=torch.bfloat16, device=weight.device)
    lo = torch.empty(shape, dtype=torch.bfloat16, device=weight.device) if split else None
    check(lib().upsnet_conv_pack_weight_bf16(stream(), ptr(weight), Cout, Cin, kh, kw, ldw, ptr(hi), ptr(lo)), "conv_pack_weight_bf16")
    return hi, lo, ldw


def conv2d_nhwc_bf16_multi(xs, whi, wlo, ldw, bias, cout, ksize, stride, pad, relu=False, residuals=None, residual_up=False,
                           out_dtype=torch.float32):
    """conv2d_nhwc_multi on the bf16 matrix cores: wlo None -> plain bf16 products, else the 3-term split (fp32-equivalent).
    residual_up (1x1 kernels): the residuals are at half resolution, added through a nearest x2 upsampling.
    Plain bf16 mode only: inputs / residuals may be torch.bfloat16 tensors (all maps of a launch alike) and out_dtype may be
    torch.bfloat16 -- bf16 activations between layers (the kernels then load / store 2-byte elements, no conversion pass)."""
    require_cuda(whi, *xs)
    assert 1 <= len(xs) <= 5
    in16 = xs[0].dtype == torch.bfloat16
    out16 = out_dtype == torch.bfloat16
    if (in16 or out16) and wlo is not None:
        raise RuntimeError("conv2d_nhwc_bf16_multi: bf16 tensors only in the plain bf16 mode")
    xs = [nhwc(x if (in16 and x.dtype == torch.bfloat16) else x.float()) for x in xs]
    if in16 and any(x.dtype != torch.bfloat16 for x in xs):
        raise RuntimeError("conv2d_nhwc_bf16_multi: mixed input dtypes in one launch")
    cin = xs[0].shape[1]
    outs, ress = [], None
    for x in xs:
        N, C, H, W = x.shape
        if C != cin:
            raise RuntimeError("conv2d_nhwc_bf16_multi: channel mismatch")
        Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
        outs.append(torch.empty((N, Ho, Wo, cout), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2) if out16
                    else _nhwc_out(N, cout, Ho, Wo, x.device))
    res16 = False
    if residuals is not None:
        res16 = residuals[0].dtype == torch.bfloat16 and wlo is None
        ress = [nhwc(r if (res16 and r.dtype == torch.bfloat16) else r.float()) for r in residuals]
        if res16 and any(r.dtype != torch.bfloat16 for r in ress):
            raise RuntimeError("conv2d_nhwc_bf16_multi: mixed residual dtypes in one launch")
        for r, o in zip(ress, outs):
            want = (o.shape[0], o.shape[1], o.shape[2] // 2, o.shape[3] // 2) if residual_up else tuple(o.shape)
            if tuple(r.shape) != want:
                raise RuntimeError("conv2d_nhwc_bf16: residual shape %s != %s" % (tuple(r.shape), want))
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_nhwc_bf16(stream(), len(xs), ptr_array(xs), ptr_array(ress) if ress is not None else None, ptr_array(outs),
                                        int_array([x.shape[0] for x in xs]), int_array([x.shape[2] for x in xs]),
                                        int_array([x.shape[3] for x in xs]), int(cin), ptr(whi), ptr(wlo), int(ldw),
                                        ptr(None if bias is None else f32c(bias)), int(cout), int(ksize), int(ksize), int(stride), int(pad),
                                        int(bool(relu)) | (2 if (residual_up and ress is not None) else 0) | (4 if in16 else 0) |
                                        (8 if out16 else 0) | (16 if res16 else 0)), "conv2d_nhwc_bf16")
    if PROFILE['enabled']:
        ev1.record()
        npix = sum(o.shape[0] * o.shape[2] * o.shape[3] for o in outs)
        nin = sum(x.shape[0] * x.shape[2] * x.shape[3] for x in xs)
        PROFILE['events'].append(('conv_bf16', ev0, ev1, 2.0 * cout * cin * ksize * ksize * npix,
                                  (2.0 if in16 else 4.0) * cin * nin + (2.0 if out16 else 4.0) * cout * npix +
                                  ((2.0 if res16 else 4.0) * cout * npix * (0.25 if residual_up else 1.0) if ress is not None else 0.0) +
                                  2.0 * cout * cin * ksize * ksize))
    return outs


def _bf16_fragments(w2d):
    """[Cout, K] fp32 -> bf16 in MFMA-fragment order [Cout/32][K/16][64 lanes][8] (lane = (k half) * 32 + row; see
    csrc/bottleneck_bf16.hip): one 16-byte load per lane and MFMA, straight from L2 into the A operand."""
    cout, k = w2d.shape
    return w2d.to(torch.bfloat16).view(cout // 32, 32, k // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous()


def pack_bottleneck_bf16(w1, w2, w3, b1, b2, b3):
    """Folded weights of an identity bottleneck ([Cm,C,1,1], [Cm,Cm,3,3], [C,Cm,1,1], biases) -> the operand pack of bottleneck_bf16."""
    require_cuda(w1, w2, w3)
    cm, c = w1.shape[0], w1.shape[1]
    if not (c == 4 * cm and tuple(w2.shape) == (cm, cm, 3, 3) and tuple(w3.shape) == (c, cm, 1, 1) and cm in (64, 128, 256, 512)):
        raise RuntimeError("pack_bottleneck_bf16: not an identity bottleneck of width 64/128/256/512")
    zeros = lambda n: torch.zeros(n, dtype=torch.float32, device=w1.device)
    return (_bf16_fragments(w1.detach().float().reshape(cm, c)),
            _bf16_fragments(w2.detach().float().permute(0, 2, 3, 1).reshape(cm, 9 * cm)),
            _bf16_fragments(w3.detach().float().reshape(c, cm)),
            zeros(cm) if b1 is None else f32c(b1.detach()), zeros(cm) if b2 is None else f32c(b2.detach()),
            zeros(c) if b3 is None else f32c(b3.detach()), cm)


def pack_bottleneck_proj_bf16(w1, w2, w3, wd, b1, b2, b3, bd):
    """Folded weights of a projection bottleneck ([Cm,Cin,1,1], [Cm,Cm,3,3], [C,Cm,1,1], projection [C,Cin,1,1], biases) -> the operand
    pack of bottleneck_proj_bf16: conv3 and the projection as one weight matrix [C, Cm + Cin], one bias b3 + bd."""
    require_cuda(w1, w2, w3, wd)
    cm, cin = w1.shape[0], w1.shape[1]
    c = 4 * cm
    if not (tuple(w2.shape) == (cm, cm, 3, 3) and tuple(w3.shape) == (c, cm, 1, 1) and tuple(wd.shape) == (c, cin, 1, 1) and
            (cm, cin) in ((64, 64), (128, 256), (256, 512))):
        raise RuntimeError("pack_bottleneck_proj_bf16: not the first bottleneck of res2 / res3 / res4")
    z = lambda b, n: torch.zeros(n, dtype=torch.float32, device=w1.device) if b is None else f32c(b.detach())
    w3d = torch.cat([w3.detach().float().reshape(c, cm), wd.detach().float().reshape(c, cin)], 1)
    return (_bf16_fragments(w1.detach().float().reshape(cm, cin)),
            _bf16_fragments(w2.detach().float().permute(0, 2, 3, 1).reshape(cm, 9 * cm)),
            _bf16_fragments(w3d), z(b1, cm), z(b2, cm), (z(b3, c) + z(bd, c)).contiguous(), cm, cin)


def bottleneck_proj_bf16(x, pack, stride):
    """relu(conv3(relu(conv2(relu(conv1_s(x))))) + proj_s(x)) of a stage's first bottleneck in one launch; x, result: bf16 channels_last."""
    w1, w2, w3d, b1, b2, b3d, cm, cin = pack
    require_cuda(x, w1)
    if x.dtype != torch.bfloat16 or x.shape[1] != cin:
        raise RuntimeError("bottleneck_proj_bf16: expects a bf16 map of %d channels" % cin)
    x = nhwc(x)
    N, C, H, W = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    out = torch.empty((N, Ho, Wo, 4 * cm), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_bottleneck_proj_bf16(stream(), ptr(x), ptr(out), N, H, W, int(cin), int(cm), int(stride), ptr(w1), ptr(w2), ptr(w3d),
                                            ptr(b1), ptr(b2), ptr(b3d)), "bottleneck_proj_bf16")
    if PROFILE['enabled']:
        ev1.record()
        k = cm * cin + 9 * cm * cm + 4 * cm * (cm + cin)
        PROFILE['events'].append(('bottleneck_bf16', ev0, ev1, 2.0 * k * N * Ho * Wo, 2.0 * C * N * Ho * Wo + 2.0 * 4 * cm * N * Ho * Wo + 2.0 * k))
    return out


def bottleneck_bf16(x, pack):
    """relu(conv3(relu(conv2(relu(conv1(x))))) + x) of an identity bottleneck in one launch; x, result: bf16 NHWC-strided NCHW."""
    w1, w2, w3, b1, b2, b3, cm = pack
    require_cuda(x, w1)
    if x.dtype != torch.bfloat16 or x.shape[1] != 4 * cm:
        raise RuntimeError("bottleneck_bf16: expects a bf16 map of %d channels" % (4 * cm))
    x = nhwc(x)
    N, C, H, W = x.shape
    out = torch.empty((N, H, W, C), dtype=torch.bfloat16, device=x.device).permute(0, 3, 1, 2)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_bottleneck_bf16(stream(), ptr(x), ptr(out), N, H, W, cm, ptr(w1), ptr(w2), ptr(w3), ptr(b1), ptr(b2), ptr(b3)),
          "bottleneck_bf16")
    if PROFILE['enabled']:
        ev1.record()
        PROFILE['events'].append(('bottleneck_bf16', ev0, ev1, 2.0 * 17 * cm * cm * N * H * W, 2.0 * 2 * C * N * H * W + 2.0 * 17 * cm * cm))
    return out


def conv2d_nhwc_splitk(x, wpack, ldw, bias, cout, ksize, stride, pad, ksplit, relu=False, residual=None):
    """conv2d_nhwc for one small map with the K walk split over `ksplit` workgroups per tile (+ a reduce/epilogue kernel)."""
    require_cuda(x, wpack)
    x = nhwc(x.float())
    N, C, H, W = x.shape
    Ho, Wo = (H + 2 * pad - ksize) // stride + 1, (W + 2 * pad - ksize) // stride + 1
    out = _nhwc_out(N, cout, Ho, Wo, x.device)
    res = None
    if residual is not None:
        res = nhwc(residual.float())
        if tuple(res.shape) != tuple(out.shape):
            raise RuntimeError("conv2d_nhwc_splitk: residual shape %s != %s" % (tuple(res.shape), tuple(out.shape)))
    ws = _ws(lib().upsnet_conv2d_splitk_workspace_bytes(N, H, W, int(cout), int(ksize), int(ksize), int(stride), int(pad), int(ksplit)), x.device)
    if PROFILE['enabled']:
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
    check(lib().upsnet_conv2d_nhwc_f32_splitk(stream(), ptr(x), ptr(res), ptr(out), N, H, W, C, ptr(wpack), int(ldw),
                                              ptr(None if bias is None else f32c(bias)), int(cout), int(ksize), int(ksize), int(stride),
                                              int(pad), int(bool(relu)), int(ksplit), ptr(ws)), "conv2d_nhwc_f32_splitk")
    if PROFILE['enabled']:
        ev1.record()
        npix = N * Ho * Wo
        PROFILE['events'].append(('conv', ev0, ev1, 2.0 * cout * C * ksize * ksize * npix,
                                  4.0 * (C * N * H * W + cout * npix * (2 if res is not None else 1) + cout * C * ksize * ksize),
                                  "split-K x%d %dx%d/%d %d->%d %s" % (ksplit, ksize, ksize, stride, C, cout, (N, H, W))))
    return out
