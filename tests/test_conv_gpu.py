"""GPU: the fp32 MFMA implicit-GEMM convolution vs torch's fp32 conv2d (reference of the same op), 1e-4."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride,pad,relu,res,bias", [
    (1, 64, 64, 33, 47, 3, 1, 1, True, False, True),
    (1, 256, 64, 40, 56, 1, 1, 0, True, False, True),
    (1, 64, 256, 40, 56, 1, 1, 0, True, True, True),
    (1, 256, 128, 41, 57, 1, 2, 0, False, False, False),
    (1, 256, 256, 32, 64, 3, 1, 1, False, False, True),
    (1, 512, 19, 24, 40, 1, 1, 0, False, False, True),
    (1, 256, 18, 17, 23, 3, 1, 1, False, False, True),
    (1, 256, 3, 16, 32, 1, 1, 0, False, False, True),
    (5, 256, 256, 14, 14, 3, 1, 1, True, False, True),
    (2, 32, 96, 9, 9, 3, 2, 1, True, True, True),
    (1, 2048, 256, 8, 16, 1, 1, 0, False, False, True),
])
def test_conv2d_nhwc_vs_torch(N, Cin, Cout, H, W, k, stride, pad, relu, res, bias):
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride, padding=pad)
    r = torch.randn_like(ref).float() if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    wp, ldw = ops.pack_conv_weight(w)
    out = ops.conv2d_nhwc(x, wp, ldw, b, Cout, k, stride, pad, relu=relu, residual=r)
    assert out.shape == ref.shape
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    # channels_last input gives the same bits as an NCHW input (layout plumbing only)
    out2 = ops.conv2d_nhwc(x.contiguous(memory_format=torch.channels_last), wp, ldw, b, Cout, k, stride, pad, relu=relu, residual=r)
    assert torch.equal(out, out2)
