"""Backward natives: HIP kernels vs the oracle vs the REFERENCE's own kernels (oracle/_ref), and the
autograd Functions end to end against a float64 torch restatement. Scatter kernels (atomics in the reference as well):
|diff| <= 1e-5 * (1 + |x|) * contributions; gather kernels (col2im_coord): bit-exact."""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from conftest import ROOT, gen_rois, torch_deform_im2col

pytestmark = pytest.mark.gpu
P = ctypes.c_void_p
CASES = [(16, 20, 33, 1, 1, 1, 1), (8, 12, 12, 2, 1, 2, 2), (4, 15, 15, 1, 2, 1, 1), (6, 9, 10, 0, 1, 1, 3)]


@pytest.fixture(scope="module")
def ref():
    path = os.path.join(ROOT, "oracle", "_ref", "libupsnet_ref.so")
    assert os.path.exists(path), "%s missing: run `make -C oracle ref` where /root/reference exists" % path
    return ctypes.CDLL(path)


_ALIVE = []


def cu(a):
    """Upload and keep alive until the next test: `p(cu(x))` must not hand a freed (and reused) block to a raw launcher."""
    t = torch.from_numpy(np.ascontiguousarray(a)).cuda()
    _ALIVE.append(t)
    return t


@pytest.fixture(autouse=True)
def _drop_uploads():
    yield
    torch.cuda.synchronize()
    del _ALIVE[:]


def p(t):
    return P(t.data_ptr())


def _geom(H, W, k, pad, stride, dil):
    return (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


def _inputs(rng, B, C, H, W, pad, stride, dil, dg, k=3):
    Ho, Wo = _geom(H, W, k, pad, stride, dil)
    im = rng.normal(size=(B, C, H, W)).astype(np.float32)
    off = (rng.normal(size=(B, dg * 2 * k * k, Ho, Wo)) * 2.5).astype(np.float32)
    mask = rng.uniform(0, 2, size=(B, dg * k * k, Ho, Wo)).astype(np.float32)
    col = rng.normal(size=(C * k * k, B, Ho, Wo)).astype(np.float32)
    return im, off, mask, col


def _close(a, b, tol=2e-5):
    np.testing.assert_allclose(a, b, rtol=tol, atol=tol)


def test_roi_align_backward(ref):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(0)
    B, C, H, W = 2, 16, 40, 64
    rois = gen_rois(rng, 200, 160, 256, 4, 150)
    rois[::3, 0] = 1
    rois = np.vstack([rois, [[0, 0, 0, 0, 0]], [[1, -30, -30, -9, -9]], [[0, 300, 3, 340, 9]]]).astype(np.float32)
    for ph, sr in ((7, 2), (14, 2), (5, 0)):
        top = rng.normal(size=(rois.shape[0], C, ph, ph)).astype(np.float32)
        want = oracle.roi_align_backward(top, rois, (B, C, H, W), 0.25, sr)
        g = torch.zeros((B, C, H, W), device='cuda')
        assert U.roi_align_backward(ph, ph, sr, 0.25, cu(top), cu(rois), g) == 1
        _close(g.cpu().numpy(), want, 1e-4)
        r = torch.zeros((B, C, H, W), device='cuda')
        ref.ref_roi_align_backward(None, p(cu(top)), ctypes.c_float(0.25), B, rois.shape[0], H, W, C, ph, ph, sr, p(cu(rois)), p(r))
        torch.cuda.synchronize()
        _close(r.cpu().numpy(), want, 1e-4)                      # the reference kernel agrees with the oracle to the same bound
        # accumulates (reference semantics): a second call doubles
        U.roi_align_backward(ph, ph, sr, 0.25, cu(top), cu(rois), g)
        _close(g.cpu().numpy(), 2 * want, 2e-4)
    assert U.roi_align_backward(7, 7, 2, 0.25, cu(top), cu(rois[:, :4]), g) == 0     # roi_align_cuda.cpp:90-93


@pytest.mark.parametrize("C,H,W,pad,stride,dil,dg", CASES)
def test_deform_col2im(ref, C, H, W, pad, stride, dil, dg):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(3)
    B, k = 2, 3
    im, off, mask, col = _inputs(rng, B, C, H, W, pad, stride, dil, dg)
    Ho, Wo = col.shape[2:]
    args = ((k, k), (pad, pad), (stride, stride), (dil, dil))
    # v1: parallel_imgs = B in one call
    want = oracle.deform_col2im(col, off, im.shape, *args, dg)
    g = torch.zeros(im.shape, device='cuda')
    U.deform_col2im(cu(col), cu(off), im.shape, col.shape, *args, B, dg, g)
    _close(g.cpu().numpy(), want)
    r = torch.zeros(im.shape, device='cuda')
    ref.ref_deform_col2im(None, p(cu(col)), p(cu(off)), C, H, W, k, k, pad, pad, stride, stride, dil, dil, B, dg, p(r))
    torch.cuda.synchronize()
    _close(r.cpu().numpy(), want)
    # v2: the pybind wrapper is batch 1 (mod_deform_conv_cuda.cpp:86), one call per image
    for b in range(B):
        cb = np.ascontiguousarray(col[:, b:b + 1])
        want = oracle.deform_col2im(cb, off[b:b + 1], (1, C, H, W), *args, dg, mask=mask[b:b + 1])
        g = torch.zeros((C, H, W), device='cuda')
        U.mod_deform_col2im(cu(cb), cu(off[b]), cu(mask[b]), (1, C, H, W), (C * k * k, Ho, Wo), *args, dg, g)
        _close(g.cpu().numpy(), want[0])
        r = torch.zeros((C, H, W), device='cuda')
        ref.ref_mod_deform_col2im(None, p(cu(cb)), p(cu(off[b])), p(cu(mask[b])), 1, C, H, W, Ho, Wo, k, k, pad, pad, stride, stride,
                                  dil, dil, dg, p(r))
        torch.cuda.synchronize()
        _close(r.cpu().numpy(), want[0])


@pytest.mark.parametrize("C,H,W,pad,stride,dil,dg", CASES)
def test_deform_col2im_coord_bit_exact(ref, C, H, W, pad, stride, dil, dg):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(4)
    B, k = 2, 3
    im, off, mask, col = _inputs(rng, B, C, H, W, pad, stride, dil, dg)
    # push some samples exactly onto / beyond the border (the inv = -2 branch and the closed/open interval ends)
    off[:, 0, 0, :] = -40.0
    off[:, 1, 1, :] = 200.0
    off[:, 2, 2, :] = np.float32(-1.0) - (np.arange(off.shape[3]) % 2)
    Ho, Wo = col.shape[2:]
    args = ((k, k), (pad, pad), (stride, stride), (dil, dil))
    want = oracle.deform_col2im_coord(col, im, off, *args, dg)
    g = torch.full(off.shape, 7.0, device='cuda')
    U.deform_col2im_coord(cu(col), cu(im), cu(off), im.shape, col.shape, *args, B, dg, g)
    assert np.array_equal(g.cpu().numpy(), want)
    r = torch.zeros(off.shape, device='cuda')
    ref.ref_deform_col2im_coord(None, p(cu(col)), p(cu(im)), p(cu(off)), C, H, W, k, k, pad, pad, stride, stride, dil, dil, B, dg, p(r))
    torch.cuda.synchronize()
    assert np.array_equal(r.cpu().numpy(), want)                 # oracle == the reference's kernel, bit for bit
    for b in range(B):
        cb = np.ascontiguousarray(col[:, b:b + 1])
        w_off, w_mask = oracle.deform_col2im_coord(cb, im[b:b + 1], off[b:b + 1], *args, dg, mask=mask[b:b + 1])
        go, gm = torch.full(off[b].shape, 7.0, device='cuda'), torch.full(mask[b].shape, 7.0, device='cuda')
        U.mod_deform_col2im_coord(cu(cb), cu(im[b]), cu(off[b]), cu(mask[b]), (1, C, H, W), (C * k * k, Ho, Wo), *args, dg, go, gm)
        assert np.array_equal(go.cpu().numpy(), w_off[0]) and np.array_equal(gm.cpu().numpy(), w_mask[0])
        ro, rm = torch.zeros(off[b].shape, device='cuda'), torch.zeros(mask[b].shape, device='cuda')
        ref.ref_mod_deform_col2im_coord(None, p(cu(cb)), p(cu(im[b])), p(cu(off[b])), p(cu(mask[b])), 1, C, H, W, Ho, Wo, k, k, pad, pad,
                                        stride, stride, dil, dil, dg, p(ro), p(rm))
        torch.cuda.synchronize()
        assert np.array_equal(ro.cpu().numpy(), w_off[0]) and np.array_equal(rm.cpu().numpy(), w_mask[0])


def test_mod_col2im_pad_quirk(ref):
    """The reference's modulated col2im launcher passes pad_h for both paddings (mod_deform_conv_kernel.cu:423): ours
    reproduces the reference's output for pad_h != pad_w."""
    from upsnet_amd import ops as U
    rng = np.random.default_rng(5)
    C, H, W, k, dg = 4, 10, 12, 3, 1
    Ho, Wo = (H + 2 * 1 - 3) + 1, (W + 2 * 2 - 3) + 1
    off = (rng.normal(size=(18, Ho, Wo)) * 2).astype(np.float32)
    mask = rng.uniform(0, 2, size=(9, Ho, Wo)).astype(np.float32)
    col = rng.normal(size=(C * 9, Ho, Wo)).astype(np.float32)
    g, r = torch.zeros((C, H, W), device='cuda'), torch.zeros((C, H, W), device='cuda')
    U.mod_deform_col2im(cu(col), cu(off), cu(mask), (1, C, H, W), (C * 9, Ho, Wo), (k, k), (1, 2), (1, 1), (1, 1), dg, g)
    ref.ref_mod_deform_col2im(None, p(cu(col)), p(cu(off)), p(cu(mask)), 1, C, H, W, Ho, Wo, k, k, 1, 2, 1, 1, 1, 1, dg, p(r))
    torch.cuda.synchronize()
    _close(g.cpu().numpy(), r.cpu().numpy())


def _torch_dcn(data, offset, mask, weight, bias, k, pad, stride, dil, dg):
    col = torch_deform_im2col(data, offset, mask, k, pad, stride, dil, dg)          # [B,C,k*k,Ho,Wo]
    B, C = data.shape[:2]
    out = torch.einsum('ok,bkp->bop', weight.reshape(weight.shape[0], -1), col.reshape(B, C * k * k, -1))
    out = out.view(B, weight.shape[0], *offset.shape[2:])
    return out if bias is None else out + bias.view(1, -1, 1, 1)


@pytest.mark.parametrize("cin,cout,dg,modulated,use_bias", [(32, 16, 1, False, True), (8, 12, 2, False, False),
                                                            (32, 8, 1, True, True), (12, 6, 3, True, False)])
def test_deform_conv_function_backward(cin, cout, dg, modulated, use_bias):
    """DeformConvFunction / ModDeformConvFunction under autograd (fused NHWC forward when Cin % 32 == 0, im2col forward
    otherwise) vs the float64 torch restatement."""
    from upsnet_amd.operators.functions.deform_conv import DeformConvFunction
    from upsnet_amd.operators.functions.mod_deform_conv import ModDeformConvFunction
    rng = np.random.default_rng(6)
    B, H, W, k, pad, stride, dil = 2, 13, 17, 3, 1, 1, 1
    Ho, Wo = _geom(H, W, k, pad, stride, dil)
    mk = lambda *s, sc=1.0: torch.from_numpy((rng.normal(size=s) * sc).astype(np.float32)).cuda()
    data, offset, weight = mk(B, cin, H, W), mk(B, dg * 18, Ho, Wo, sc=2.0), mk(cout, cin, k, k, sc=0.1)
    mask = torch.from_numpy(rng.uniform(0, 2, size=(B, dg * 9, Ho, Wo)).astype(np.float32)).cuda() if modulated else None
    bias = mk(cout) if use_bias else None
    G = mk(B, cout, Ho, Wo)
    leaves = [t for t in (data, offset, mask, weight, bias) if t is not None]
    for t in leaves:
        t.requires_grad_()
    if modulated:
        out = ModDeformConvFunction.apply(data, offset, mask, weight, bias, cin, cout, (k, k), (stride,) * 2, (pad,) * 2, (dil,) * 2, 1, dg)
    else:
        out = DeformConvFunction.apply(data, offset, weight, bias, cin, cout, (k, k), (stride,) * 2, (pad,) * 2, (dil,) * 2, 1, dg)
    (out * G).sum().backward()
    d64 = [t.detach().double().cpu().requires_grad_() for t in leaves]
    it = iter(d64)
    r_data, r_off = next(it), next(it)
    r_mask = next(it) if modulated else None
    r_w = next(it)
    r_b = next(it) if use_bias else None
    want = _torch_dcn(r_data, r_off, r_mask, r_w, r_b, k, pad, stride, dil, dg)
    np.testing.assert_allclose(out.detach().cpu().numpy(), want.detach().numpy(), rtol=1e-4, atol=1e-4)
    (want * G.double().cpu()).sum().backward()
    for t, r in zip(leaves, d64):
        assert t.grad is not None and t.grad.shape == t.shape
        np.testing.assert_allclose(t.grad.cpu().numpy(), r.grad.numpy(), rtol=2e-4, atol=2e-4)


def test_roi_align_function_backward():
    """RoIAlignFunction(...)(features, rois) is differentiable w.r.t. the features, and the legacy .backward(grad) call
    of the reference (functions/roialign.py:45-54) returns the same gradient."""
    from upsnet_amd.operators.functions.roialign import RoIAlignFunction
    rng = np.random.default_rng(7)
    feat = torch.from_numpy(rng.normal(size=(2, 8, 24, 40)).astype(np.float32)).cuda().requires_grad_()
    rois = gen_rois(rng, 50, 96, 160, 4, 90).astype(np.float32)
    rois[::2, 0] = 1
    top = rng.normal(size=(50, 8, 7, 7)).astype(np.float32)
    fn = RoIAlignFunction(7, 7, 0.25, 2)
    out = fn(feat, cu(rois))
    assert np.array_equal(out.detach().cpu().numpy(), oracle.roi_align_forward(feat.detach().cpu().numpy(), rois, 7, 7, 0.25, 2))
    (out * cu(top)).sum().backward()
    want = oracle.roi_align_backward(top, rois, feat.shape, 0.25, 2)
    _close(feat.grad.cpu().numpy(), want, 1e-4)
    g, none = fn.backward(cu(top))
    assert none is None
    _close(g.cpu().numpy(), want, 1e-4)
