"""Pins the oracle (and the HIP kernels) against the REFERENCE's own GPU kernels.

oracle/_ref/libupsnet_ref.so is built by oracle/Makefile straight from /root/reference (hipify-perl +
hipcc, -ffp-contract=off) and shipped to the GPU box prebuilt; this file only loads it. If the library
is absent the tests fail (not skip): the pin is part of the parity claim.
"""
import ctypes
import os

import numpy as np
import pytest
import torch

import oracle
from oracle import ops as oops
from conftest import ROOT, gen_dets, gen_rois

pytestmark = pytest.mark.gpu
P = ctypes.c_void_p


def _load(name):
    path = os.path.join(ROOT, "oracle", "_ref", name)
    assert os.path.exists(path), "%s missing: run `make -C oracle ref` where /root/reference exists" % path
    return ctypes.CDLL(path)


@pytest.fixture(scope="module")
def ref():
    return _load("libupsnet_ref.so")


@pytest.fixture(scope="module")
def ref_fma():
    return _load("libupsnet_ref_fma.so")


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def p(t):
    return P(t.data_ptr())


def _ref_roi(lib, feat, rois, ph, scale):
    f, r = cu(feat), cu(rois)
    out = torch.zeros((rois.shape[0], feat.shape[1], ph, ph), device='cuda')
    lib.ref_roi_align_forward(None, p(f), ctypes.c_float(scale), rois.shape[0], feat.shape[2], feat.shape[3], feat.shape[1],
                              ph, ph, 2, p(r), p(out))
    torch.cuda.synchronize()
    return out.cpu().numpy()


def test_reference_roi_align(ref, ref_fma):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(0)
    feat = rng.normal(size=(1, 16, 40, 64)).astype(np.float32)
    rois = gen_rois(rng, 200, 160, 256, 4, 150)
    rois = np.vstack([rois, [[0, 0, 0, 0, 0]], [[0, -30, -30, -9, -9]], [[0, 300, 3, 340, 9]]]).astype(np.float32)
    for ph in (7, 14):
        r = _ref_roi(ref, feat, rois, ph, 0.25)
        assert np.array_equal(r, oracle.roi_align_forward(feat, rois, ph, ph, 0.25))          # oracle == reference kernel
        assert np.array_equal(r, U.roi_align_nchw(cu(feat), cu(rois), ph, ph, 0.25).cpu().numpy())  # ours == reference kernel
        np.testing.assert_allclose(_ref_roi(ref_fma, feat, rois, ph, 0.25), r, rtol=0, atol=1e-4)  # FMA build (nvcc default): within the 1e-4 logit tolerance


def _ref_im2col(lib, im, off, k, pad, stride, dil, dg, mask=None):
    C, H, W = im.shape
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    col = torch.zeros((C * k * k, Ho, Wo), device='cuda')
    i, o = cu(im), cu(off)
    if mask is None:
        lib.ref_deform_im2col(None, p(i), p(o), C, H, W, k, k, pad, pad, stride, stride, dil, dil, 1, dg, p(col))
    else:
        m = cu(mask)
        lib.ref_mod_deform_im2col(None, p(i), p(o), p(m), 1, C, H, W, Ho, Wo, k, k, pad, pad, stride, stride, dil, dil, dg, p(col))
    torch.cuda.synchronize()
    return col.cpu().numpy()


@pytest.mark.parametrize("C,H,W,pad,stride,dil,dg", [(16, 20, 33, 1, 1, 1, 1), (8, 12, 12, 2, 1, 2, 2), (4, 15, 15, 1, 2, 1, 1)])
def test_reference_deform_im2col(ref, ref_fma, C, H, W, pad, stride, dil, dg):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(1)
    k = 3
    im = rng.normal(size=(C, H, W)).astype(np.float32)
    Ho, Wo = U.out_hw(H, W, (k, k), (pad, pad), (stride, stride), (dil, dil))
    off = (rng.normal(size=(dg * 18, Ho, Wo)) * 2.5).astype(np.float32)
    mask = rng.uniform(0, 2, size=(dg * 9, Ho, Wo)).astype(np.float32)
    for mk in (None, mask):
        r = _ref_im2col(ref, im, off, k, pad, stride, dil, dg, mk)
        assert np.array_equal(r, oracle.deform_im2col(im, off, (k, k), (pad, pad), (stride, stride), (dil, dil), dg, mask=mk))
        col = torch.zeros(r.shape, device='cuda')
        if mk is None:
            U.deform_im2col(cu(im), cu(off), (1, C, H, W), r.shape, (k, k), (pad, pad), (stride, stride), (dil, dil), 1, dg, col)
        else:
            U.mod_deform_im2col(cu(im), cu(off), cu(mk), (1, C, H, W), r.shape, (k, k), (pad, pad), (stride, stride), (dil, dil), dg, col)
        assert np.array_equal(col.cpu().numpy(), r)
        np.testing.assert_allclose(_ref_im2col(ref_fma, im, off, k, pad, stride, dil, dg, mk), r, rtol=0, atol=1e-4)


@pytest.mark.parametrize("n", [1, 64, 65, 500, 1000, 2000])
def test_reference_nms(ref, n):
    from upsnet_amd import ops as U
    rng = np.random.default_rng(n)
    d = gen_dets(rng, n)
    order = oops.argsort_desc(d[:, 4])
    sd = np.ascontiguousarray(d[order])
    keep = np.zeros((n,), np.int32)
    num = ctypes.c_int(0)
    ref.ref_nms(keep.ctypes.data_as(P), ctypes.byref(num), sd.ctypes.data_as(P), n, 5, ctypes.c_float(0.6), 0)
    rk = keep[:num.value]
    assert np.array_equal(rk, oracle.nms_sorted(sd, 0.6))                 # oracle == the reference's _nms
    assert np.array_equal(rk, U.nms_host(sd, 0.6))                        # our `_nms` drop-in == reference
    assert np.array_equal(order[rk], U.gpu_nms(cu(d), 0.6).cpu().numpy())  # device pipeline == gpu_nms(order[keep])
