import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


# ---------------------------------------------------------------- shared synthetic generators
def gen_rois(rng, n, im_h=1024, im_w=2048, smin=16, smax=512):
    size = np.exp(rng.uniform(np.log(smin), np.log(smax), n))
    ar = np.exp(rng.uniform(np.log(0.5), np.log(2.0), n))
    w, h = size * np.sqrt(ar), size / np.sqrt(ar)
    cx, cy = rng.uniform(0, im_w, n), rng.uniform(0, im_h, n)
    b = np.stack([cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2], 1)
    b[:, 0::2] = np.clip(b[:, 0::2], 0, im_w - 1)
    b[:, 1::2] = np.clip(b[:, 1::2], 0, im_h - 1)
    return np.hstack([np.zeros((n, 1)), b]).astype(np.float32)


def gen_dets(rng, n, im_h=1024, im_w=2048, ties=True, cluster=True):
    r = gen_rois(rng, n, im_h, im_w)[:, 1:]
    if cluster and n > 8:  # jittered copies so that NMS actually suppresses
        src = rng.integers(0, n, n // 2)
        r[n // 2:n // 2 + len(src)] = r[src] + rng.normal(0, 4, (len(src), 4)).astype(np.float32)
    s = rng.uniform(0, 1, n).astype(np.float32)
    if ties and n > 4:
        k = max(1, n // 20)
        s[rng.integers(0, n, k)] = s[rng.integers(0, n, 1)]
        s[rng.integers(0, n, k)] = np.float32(1.0)  # saturated scores
    return np.hstack([r, s[:, None]]).astype(np.float32)


def gen_panoptic_maps(rng, H, W, k, num_stuff=11, num_things=8, void_frac=0.03, small_stuff=True):
    """Synthetic (seg, pan, cls_ind) like the network's outputs: stuff stripes, k rectangular instances (ids num_stuff+j) whose
    semantic votes are a mix of: own thing class majority / a stuff class majority >= 50 % / a split vote; some void; one tiny
    stuff area (below stuff_area_limit) when small_stuff."""
    seg = np.zeros((H, W), np.int64)
    bands = np.sort(rng.choice(np.arange(8, W - 8), size=num_stuff - 1, replace=False))
    for c, (a, b) in enumerate(zip(np.r_[0, bands], np.r_[bands, W])):
        seg[:, a:b] = c
    if small_stuff:
        seg[:, :] = np.where(seg == num_stuff - 1, 0, seg)
        seg[0:3, 0:5] = num_stuff - 1
    pan = seg.copy()
    cls_ind = rng.integers(1, num_things + 1, size=k).astype(np.int64)
    for j in range(k):
        h, w = int(rng.integers(6, H // 3)), int(rng.integers(6, W // 4))
        y, x = int(rng.integers(0, H - h)), int(rng.integers(0, W - w))
        pan[y:y + h, x:x + w] = num_stuff + j
        mode = j % 4
        thing = cls_ind[j] + num_stuff - 1
        if mode == 0:        # agrees
            seg[y:y + h, x:x + w] = thing
        elif mode == 1:      # stuff majority >= 50 %
            seg[y:y + h, x:x + w] = int(rng.integers(0, num_stuff - 1))
            seg[y:y + h // 3, x:x + w] = thing
        elif mode == 2:      # another thing class wins -> keeps the instance class
            seg[y:y + h, x:x + w] = (cls_ind[j] % num_things) + num_stuff
        else:                # split three ways, stuff plurality below 50 %
            seg[y:y + h, x:x + w] = thing
            seg[y:y + h, x:x + 2 * w // 5] = 1
            seg[y:y + h, x + 2 * w // 5:x + 7 * w // 10] = (cls_ind[j] % num_things) + num_stuff
    void = rng.random((H, W)) < void_frac
    pan[void] = 255
    return seg, pan, cls_ind


def torch_deform_im2col(im, off, mask, k, pad, stride, dil, dg):
    """Differentiable float64 deformable im2col (zero outside the image, bilinear inside): [B,C,k*k,Ho,Wo]."""
    B, C, H, W = im.shape
    Ho, Wo = off.shape[2:]
    ys = torch.arange(Ho, dtype=torch.float64).view(1, 1, 1, Ho, 1) * stride - pad
    xs = torch.arange(Wo, dtype=torch.float64).view(1, 1, 1, 1, Wo) * stride - pad
    ki = (torch.arange(k * k) // k).view(1, 1, k * k, 1, 1).double() * dil
    kj = (torch.arange(k * k) % k).view(1, 1, k * k, 1, 1).double() * dil
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
