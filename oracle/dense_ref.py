"""oracle.dense_ref -- TEST INFRASTRUCTURE ONLY.

An independent float64 execution of the DENSE part of the reference's inference graph, written as plain torch functional
calls that read the parameters straight out of a model's modules (by the reference's attribute names = its state-dict keys).
It does not call any module's forward(), nothing from upsnet_amd.models.hipconv and no HIP kernel, so it checks what the
per-op parity tests cannot: the WIRING of the hand-written convolution path (dispatch, weight packing caches, fused
epilogues, folded BN, multi-map launches, the commuted score tail, the fc6 re-layout) against the graph the reference
defines:

    backbone   upsnet/models/resnet.py:53-100 (Bottleneck), :102-153 (DCNBottleneck), :155-175 (conv1 stem), :347-356
    FPN        upsnet/models/fpn.py:78-104
    RPN        upsnet/models/rpn.py:52-57
    FCN head   upsnet/models/fcn.py:29-58 (subnet), :88-108 (head)
    box head   upsnet/models/rcnn.py:132-146, ROI pooling upsnet/operators/modules/fpn_roi_align.py:32-62
    mask head  upsnet/models/rcnn.py:79-87
    deformable convolution   upsnet/operators/src/deform_conv_kernel.cu:88-118,194-242 (bilinear sampling, zero outside)
    ROIAlign                 upsnet/operators/src/roi_align_kernel.cu:43-95,163-235 (sampling_ratio 2, legacy alignment)

Everything is evaluated in float64 (device of the model's parameters; torch's native double convolution), so the result is
the "true" value both fp32 executions -- the reference's cuDNN one and ours -- approximate; the tests require the product's
fp32 logits within rtol = atol = 1e-4 of it (north_star: "fp32 logits within 1e-4").
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops as oops

D = torch.float64   # evaluation dtype; dense_reference(..., dtype=torch.float32) re-runs the same graph as a plain fp32 library execution


def _w(t):
    return None if t is None else t.detach().to(D)


def conv_bn(x, conv, bn=None, relu=False):
    """nn.Conv2d followed by an (eval-mode) BatchNorm2d that may have been folded away (nn.Identity) or be absent."""
    y = F.conv2d(x, _w(conv.weight), _w(conv.bias), conv.stride, conv.padding, conv.dilation, conv.groups)
    if isinstance(bn, nn.BatchNorm2d):
        scale = _w(bn.weight) / torch.sqrt(_w(bn.running_var) + bn.eps)
        y = y * scale.view(1, -1, 1, 1) + (_w(bn.bias) - _w(bn.running_mean) * scale).view(1, -1, 1, 1)
    return F.relu(y) if relu else y


def deform_im2col(im, off, mask, k, pad, stride, dil, dg):
    """Differentiable float64 deformable im2col (zero outside the image, bilinear inside): [B,C,k*k,Ho,Wo].
    deform_conv_kernel.cu:88-118 (bilinear), :227-240 (sampling positions); mask = DCN v2 modulation."""
    B, C, H, W = im.shape
    Ho, Wo = off.shape[2:]
    dev = im.device
    ys = torch.arange(Ho, dtype=D, device=dev).view(1, 1, 1, Ho, 1) * stride - pad
    xs = torch.arange(Wo, dtype=D, device=dev).view(1, 1, 1, 1, Wo) * stride - pad
    ki = (torch.arange(k * k, device=dev) // k).view(1, 1, k * k, 1, 1).to(D) * dil
    kj = (torch.arange(k * k, device=dev) % k).view(1, 1, k * k, 1, 1).to(D) * dil
    o = off.view(B, dg, k * k, 2, Ho, Wo)
    ph, pw = ys + ki + o[:, :, :, 0], xs + kj + o[:, :, :, 1]                     # [B,dg,k*k,Ho,Wo]
    cpg = C // dg
    ph, pw = ph.repeat_interleave(cpg, 1), pw.repeat_interleave(cpg, 1)          # [B,C,k*k,Ho,Wo]
    h0, w0 = torch.floor(ph).detach(), torch.floor(pw).detach()
    flat = im.reshape(B, C, H * W)
    val = 0
    for dy in (0, 1):
        for dx in (0, 1):
            hh, ww = h0 + dy, w0 + dx
            ok = (hh >= 0) & (hh <= H - 1) & (ww >= 0) & (ww <= W - 1)
            idx = (hh.clamp(0, H - 1) * W + ww.clamp(0, W - 1)).long().view(B, C, -1)
            v = torch.gather(flat, 2, idx).view_as(ph) * ok
            wy = (ph - h0) if dy else (1 - (ph - h0))
            wx = (pw - w0) if dx else (1 - (pw - w0))
            val = val + wy * wx * v
    if mask is not None:
        val = val * mask.view(B, dg, k * k, Ho, Wo).repeat_interleave(cpg, 1)
    return val


def deform_conv(x, offset, dc, bn=None, relu=False, chunk=64):
    """DeformConv (functions/deform_conv.py:43-57): im2col + GEMM + bias; channel-chunked so that the column tensor stays small."""
    B, C, H, W = x.shape
    k = dc.kernel_size[0]
    w = _w(dc.weight)                                   # [Cout, Cin, k, k]
    out = None
    for c0 in range(0, C, chunk):
        c1 = min(C, c0 + chunk)
        assert dc.deformable_groups == 1
        col = deform_im2col(x[:, c0:c1], offset, None, k, dc.padding[0], dc.stride[0], dc.dilation[0], 1)   # [B,c,k*k,Ho,Wo]
        part = torch.einsum('ock,bckhw->bohw', w[:, c0:c1].reshape(w.shape[0], c1 - c0, k * k), col)
        out = part if out is None else out + part
    if dc.bias is not None:
        out = out + _w(dc.bias).view(1, -1, 1, 1)
    if isinstance(bn, nn.BatchNorm2d):
        scale = _w(bn.weight) / torch.sqrt(_w(bn.running_var) + bn.eps)
        out = out * scale.view(1, -1, 1, 1) + (_w(bn.bias) - _w(bn.running_mean) * scale).view(1, -1, 1, 1)
    return F.relu(out) if relu else out


# ----------------------------------------------------------------------------- backbone / FPN / RPN
def bottleneck(x, blk, given=None, own=None):
    """resnet.py:84-100 (Bottleneck.forward) / :133-153 (DCNBottleneck.forward). given: iterator over recorded sampling offsets
    (one per deformable block, in graph order) to sample at instead of the ones predicted here; own: list receiving the latter."""
    y = conv_bn(x, blk.conv1, blk.bn1, relu=True)
    if hasattr(blk, 'conv2_offset'):
        off = conv_bn(y, blk.conv2_offset)
        if own is not None:
            own.append(off)
        if given is not None:
            off = next(given).to(device=y.device, dtype=y.dtype)
        y = deform_conv(y, off, blk.conv2, blk.bn2, relu=True)
    else:
        y = conv_bn(y, blk.conv2, blk.bn2, relu=True)
    y = conv_bn(y, blk.conv3, blk.bn3)
    sc = x if blk.downsample is None else conv_bn(x, blk.downsample[0], blk.downsample[1])
    return F.relu(y + sc)


def backbone(x, bb, given_offsets=None, own_offsets=None):
    """resnet.py:347-356: conv1 (7x7/2 + BN + ReLU + 3x3/2 max-pool, :169-175) then res2..res5. given_offsets: recorded sampling
    offsets of the deformable bottlenecks in graph order (a chaotic chain of 30 data-dependent samplers cannot be compared
    free-running: the strict check is made at identical sampling positions, and the offset predictions on their own)."""
    given = iter(given_offsets) if given_offsets is not None else None
    y = conv_bn(x, bb.conv1.conv1, bb.conv1.bn1, relu=True)
    y = F.max_pool2d(y, kernel_size=3, stride=2, padding=1)
    feats = []
    for name in ('res2', 'res3', 'res4', 'res5'):
        for blk in getattr(bb, name).layers:
            y = bottleneck(y, blk, given, own_offsets)
        feats.append(y)
    return feats


def fpn(res2, res3, res4, res5, m):
    """fpn.py:78-104 (nearest x2 top-down; GAP branch :84-86 when the module has fpn_gap)."""
    p5_1 = conv_bn(res5, m.fpn_p5_1x1)
    if hasattr(m, 'fpn_gap'):
        gap = F.linear(F.adaptive_avg_pool2d(res5, (1, 1)).flatten(1), _w(m.fpn_gap.weight), _w(m.fpn_gap.bias))
        p5_1 = p5_1 + gap.view(-1, m.feature_dim, 1, 1)

    def up(t):
        return F.interpolate(t, scale_factor=2, mode='nearest')
    p4_plus = up(p5_1) + conv_bn(res4, m.fpn_p4_1x1)
    p3_plus = up(p4_plus) + conv_bn(res3, m.fpn_p3_1x1)
    p2_plus = up(p3_plus) + conv_bn(res2, m.fpn_p2_1x1)
    p5, p4, p3, p2 = conv_bn(p5_1, m.fpn_p5), conv_bn(p4_plus, m.fpn_p4), conv_bn(p3_plus, m.fpn_p3), conv_bn(p2_plus, m.fpn_p2)
    p6 = F.max_pool2d(p5, kernel_size=1, stride=2)
    return [p2, p3, p4, p5, p6]


def rpn(feat, m):
    """rpn.py:52-57 -> (cls_score, bbox_pred, cls_prob)."""
    x = conv_bn(feat, m.conv_proposal[0], relu=True)
    score = conv_bn(x, m.cls_score)
    return score, conv_bn(x, m.bbox_pred), torch.sigmoid(score)


# ----------------------------------------------------------------------------- semantic head
def fcn_score(feats, head, given_offsets=None, own_offsets=None):
    """fcn.py:88-108 up to the class scores at 1/4 resolution (the product's `fcn_score`; fcn_output = its bilinear x4).
    given_offsets[layer][level]: sample at THESE offsets instead of the ones predicted here (a deformable layer amplifies an
    offset difference d by the local feature gradient: |d| ~ 1e-5 px on features of magnitude ~100 moves a sample by ~1e-3, so
    the strict 1e-4 check of the sampling + GEMM stages is made at identical sampling positions and the offset predictions are
    checked on their own); own_offsets (a list) receives the offsets predicted here, [layer][level]."""
    lv = []
    for l, f in enumerate(feats[:4]):
        y = f
        for i in range(head.fcn_subnet.num_layers):     # fcn.py:51-58: DeformConvWithOffset + ReLU
            layer = head.fcn_subnet.conv[i][0]
            off = conv_bn(y, layer.conv_offset)
            if own_offsets is not None:
                while len(own_offsets) <= i:
                    own_offsets.append([None] * 4)
                own_offsets[i][l] = off
            if given_offsets is not None:
                off = given_offsets[i][l].to(device=y.device, dtype=y.dtype)
            y = deform_conv(y, off, layer.conv, relu=True)
        lv.append(y)
    for l, s in ((1, 2), (2, 4), (3, 8)):
        lv[l] = F.interpolate(lv[l], None, s, mode='bilinear', align_corners=False)
    return conv_bn(torch.cat(lv, 1), head.score)


# ----------------------------------------------------------------------------- ROI heads
def roi_align(feat, rois, ph, pw, scale, sampling=2):
    """roi_align_kernel.cu:43-95,163-235 in float64: feat [1,C,H,W], rois [n,5] (batch index ignored: one image) -> [n,C,ph,pw]."""
    n = rois.shape[0]
    C, H, W = feat.shape[1:]
    dev = feat.device
    r = rois.to(D)
    x1, y1, x2, y2 = (r[:, i] * scale for i in (1, 2, 3, 4))
    rw, rh = torch.clamp(x2 - x1, min=1.0), torch.clamp(y2 - y1, min=1.0)
    bw, bh = rw / pw, rh / ph
    g = sampling
    iy = (torch.arange(ph * g, device=dev, dtype=D) // g).view(1, -1) * bh.view(-1, 1) + \
         ((torch.arange(ph * g, device=dev, dtype=D) % g) + 0.5).view(1, -1) * bh.view(-1, 1) / g + y1.view(-1, 1)     # [n, ph*g]
    ix = (torch.arange(pw * g, device=dev, dtype=D) // g).view(1, -1) * bw.view(-1, 1) + \
         ((torch.arange(pw * g, device=dev, dtype=D) % g) + 0.5).view(1, -1) * bw.view(-1, 1) / g + x1.view(-1, 1)     # [n, pw*g]

    def prep(v, size):
        empty = (v < -1.0) | (v > size)
        v = torch.clamp(v, min=0.0)
        lo = torch.floor(v).long()
        at_edge = lo >= size - 1
        lo = torch.where(at_edge, torch.full_like(lo, size - 1), lo)
        hi = torch.where(at_edge, lo, lo + 1)
        v = torch.where(at_edge, lo.to(D), v)
        frac = v - lo.to(D)
        return lo, hi, frac, empty
    ylo, yhi, fy, ey = prep(iy, H)
    xlo, xhi, fx, ex = prep(ix, W)
    flat = feat[0].reshape(C, H * W)

    def tap(yy, xx):                                     # [n,Py] x [n,Px] -> [n,C,Py,Px]
        idx = (yy.unsqueeze(2) * W + xx.unsqueeze(1)).reshape(-1)
        return flat[:, idx].view(C, n, yy.shape[1], xx.shape[1]).permute(1, 0, 2, 3)
    hy, ly, hx, lx = (1 - fy).view(n, 1, -1, 1), fy.view(n, 1, -1, 1), (1 - fx).view(n, 1, 1, -1), fx.view(n, 1, 1, -1)
    val = hy * hx * tap(ylo, xlo) + hy * lx * tap(ylo, xhi) + ly * hx * tap(yhi, xlo) + ly * lx * tap(yhi, xhi)
    val = val * (~(ey.view(n, 1, -1, 1) | ex.view(n, 1, 1, -1))).to(D)
    return val.view(n, C, ph, g, pw, g).sum(dim=(3, 5)) / (g * g)


def fpn_roi_pool(feats, rois, size, chunk=128):
    """fpn_roi_align.py:32-62: level by sqrt(area) (same fp32 rule as the product: oracle.ops.fpn_level), pooled per level."""
    lvl = oops.fpn_level(rois.detach().cpu().numpy().astype(np.float32))
    out = torch.zeros((rois.shape[0], feats[0].shape[1], size, size), dtype=D, device=feats[0].device)
    for l in range(4):
        idx = np.where(lvl == l)[0]
        for c0 in range(0, len(idx), chunk):
            sel = torch.as_tensor(idx[c0:c0 + chunk], device=rois.device)
            out[sel] = roi_align(feats[l], rois[sel], size, size, 1.0 / (4 << l))
    return out


def box_head(feats, rois, m):
    """rcnn.py:132-146 -> (cls_prob, bbox_pred)."""
    pool = fpn_roi_pool(feats, rois, m.pool_size)
    fc6 = F.relu(F.linear(pool.reshape(pool.shape[0], -1), _w(m.fc6[0].weight), _w(m.fc6[0].bias)))
    fc7 = F.relu(F.linear(fc6, _w(m.fc7[0].weight), _w(m.fc7[0].bias)))
    cls_score = F.linear(fc7, _w(m.cls_score.weight), _w(m.cls_score.bias))
    return F.softmax(cls_score, dim=1), F.linear(fc7, _w(m.bbox_pred.weight), _w(m.bbox_pred.bias))


def mask_head(feats, boxes, m, size):
    """rcnn.py:79-87 -> mask logits [n, num_classes, 2*size, 2*size]."""
    x = fpn_roi_pool(feats, boxes, size)
    for blk in (m.mask_conv1, m.mask_conv2, m.mask_conv3, m.mask_conv4):
        x = conv_bn(x, blk[0], relu=True)
    d = m.mask_deconv1[0]
    x = F.relu(F.conv_transpose2d(x, _w(d.weight), _w(d.bias), d.stride, d.padding))
    return conv_bn(x, m.mask_score)


# ----------------------------------------------------------------------------- driver
def dense_reference(model, data, rois, det_boxes, pan_boxes, mask_size, dtype=torch.float64, fcn_offsets=None, backbone_offsets=None):
    """The dense stages of resnet_upsnet.forward (resnet_upsnet.py:88-248, test branch) in float64, with the SELECTION results
    (rois, detections) taken from the caller (the product's recorded ones: selection is integer work with its own bit-exact
    tests, and tiny logit differences must not be allowed to change which boxes the two executions look at).
    fcn_offsets: see fcn_score(given_offsets); the result then also holds 'fcn_score_given' and 'fcn_offsets' (own predictions
    along the given-offset chain)."""
    global D
    saved, D = D, dtype
    try:
        return _dense_reference(model, data, rois, det_boxes, pan_boxes, mask_size, fcn_offsets, backbone_offsets)
    finally:
        D = saved


def _dense_reference(model, data, rois, det_boxes, pan_boxes, mask_size, fcn_offsets, backbone_offsets=None):
    with torch.no_grad():
        dev = next(model.parameters()).device
        x = data['data'].to(dev).to(D)
        if x.shape[1] == 4:
            x = x[:, :3]
        own_bb = []
        res = backbone(x.contiguous(), model.resnet_backbone, backbone_offsets, own_bb)
        pyr = fpn(*res, model.fpn)
        r = [rpn(f, model.rpn) for f in pyr]
        out = dict(res=res, pyramid=pyr, rpn_cls_prob=[t[2] for t in r], rpn_bbox_pred=[t[1] for t in r], backbone_offsets=own_bb)
        out['fcn_score'] = fcn_score(pyr, model.fcn_head)
        if fcn_offsets is not None:
            own = []
            out['fcn_score_given'] = fcn_score(pyr, model.fcn_head, given_offsets=fcn_offsets, own_offsets=own)
            out['fcn_offsets'] = own
        out['cls_prob'], out['bbox_pred'] = box_head(pyr, rois.to(dev), model.rcnn)
        out['mask_logit_det'] = mask_head(pyr, det_boxes.to(dev), model.mask_branch, mask_size // 2)
        out['mask_logit_pan'] = mask_head(pyr, pan_boxes.to(dev), model.mask_branch, mask_size // 2)
    return out
