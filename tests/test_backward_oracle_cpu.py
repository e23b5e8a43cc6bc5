"""Pins the oracle's backward restatements (oracle/c: orc_roi_align_backward, orc_deform_col2im, orc_deform_col2im_coord)
without a GPU: the scatter ops are the exact adjoints of forward ops that are themselves pinned against the reference
kernels (tests/test_ref_kernels_gpu.py), and the coordinate gradients are checked against torch autograd in float64."""
import numpy as np
import pytest
import torch

import oracle
from conftest import gen_rois, torch_deform_im2col as _torch_im2col

CASES = [(8, 11, 14, 1, 1, 1, 1), (8, 12, 12, 2, 1, 2, 2), (4, 15, 15, 1, 2, 1, 1), (6, 9, 10, 0, 1, 1, 3)]


def _geom(H, W, k, pad, stride, dil):
    return (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _inputs(rng, B, C, H, W, pad, stride, dil, dg, k=3):
    Ho, Wo = _geom(H, W, k, pad, stride, dil)
    im = rng.normal(size=(B, C, H, W)).astype(np.float32)
    off = (rng.normal(size=(B, dg * 2 * k * k, Ho, Wo)) * 2.5).astype(np.float32)
    mask = rng.uniform(0, 2, size=(B, dg * k * k, Ho, Wo)).astype(np.float32)
    col = rng.normal(size=(C * k * k, B, Ho, Wo)).astype(np.float32)
    return im, off, mask, col


def test_roi_align_backward_is_adjoint_of_forward():
    rng = np.random.default_rng(0)
    feat = rng.normal(size=(2, 6, 20, 32)).astype(np.float32)
    rois = gen_rois(rng, 60, 80, 128, 4, 100)
    rois[::2, 0] = 1
    rois = np.vstack([rois, [[0, 0, 0, 0, 0]], [[1, -30, -30, -9, -9]], [[0, 120, 3, 140, 9]]]).astype(np.float32)
    for ph, sr in ((7, 2), (14, 2), (3, 0)):
        top = rng.normal(size=(rois.shape[0], 6, ph, ph)).astype(np.float32)
        # <forward(feat), top> == <feat, backward(top)> (both sides accumulated in float64)
        out = oracle.roi_align_forward(feat, rois, ph, ph, 0.25, sr)
        g = oracle.roi_align_backward(top, rois, feat.shape, 0.25, sr)
        lhs = float((out.astype(np.float64) * top).sum())
        rhs = float((feat.astype(np.float64) * g).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)
        assert np.abs(g).sum() > 0


@pytest.mark.parametrize("C,H,W,pad,stride,dil,dg", CASES)
def test_col2im_is_adjoint_of_im2col(C, H, W, pad, stride, dil, dg):
    rng = np.random.default_rng(1)
    B = 2
    im, off, mask, col = _inputs(rng, B, C, H, W, pad, stride, dil, dg)
    args = ((3, 3), (pad, pad), (stride, stride), (dil, dil), dg)
    for mk in (None, mask):
        g = oracle.deform_col2im(col, off, im.shape, *args, mask=mk)
        lhs = 0.0
        for b in range(B):
            fwd = oracle.deform_im2col(im[b], off[b], *args, mask=None if mk is None else mk[b])
            lhs += float((fwd.astype(np.float64) * col[:, b]).sum())
        rhs = float((im.astype(np.float64) * g).sum())
        assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs)), (lhs, rhs)


@pytest.mark.parametrize("C,H,W,pad,stride,dil,dg", CASES)
def test_col2im_coord_matches_autograd(C, H, W, pad, stride, dil, dg):
    rng = np.random.default_rng(2)
    B = 2
    im, off, mask, col = _inputs(rng, B, C, H, W, pad, stride, dil, dg)
    args = ((3, 3), (pad, pad), (stride, stride), (dil, dil), dg)
    for mk in (None, mask):
        t_im = torch.from_numpy(im).double().requires_grad_()
        t_off = torch.from_numpy(off).double().requires_grad_()
        t_mask = None if mk is None else torch.from_numpy(mk).double().requires_grad_()
        val = _torch_im2col(t_im, t_off, t_mask, 3, pad, stride, dil, dg)       # [B,C,9,Ho,Wo]
        G = torch.from_numpy(col).double().view(C, 9, B, *col.shape[2:]).permute(2, 0, 1, 3, 4)
        (val * G).sum().backward()
        res = oracle.deform_col2im_coord(col, im, off, *args, mask=mk)
        g_off = res if mk is None else res[0]
        np.testing.assert_allclose(g_off, t_off.grad.numpy(), rtol=1e-4, atol=1e-4)
        if mk is not None:
            np.testing.assert_allclose(res[1], t_mask.grad.numpy(), rtol=1e-4, atol=1e-4)
        # and the image gradient of the same loss is col2im
        np.testing.assert_allclose(oracle.deform_col2im(col, off, im.shape, *args, mask=mk), t_im.grad.numpy(), rtol=1e-4, atol=1e-4)
