"""GPU: the one-launch stem (conv 7x7/2/3 + bias + ReLU + max-pool 3x3/2/1; upsnet/models/resnet.py:347-356): csrc/stem_pool.hip (fp32
MFMA, the headline path) vs float64 at 1e-4; csrc/stem_pool_bf16.hip (bf16 mode) vs float64 on the bf16-rounded operands with the
convolution result rounded to bf16 before the pool."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,bias", [(1, 64, 128, True), (2, 37, 53, True), (1, 200, 333, False), (1, 9, 7, True), (1, 130, 66, True)])
def test_stem_pool_bf16_vs_float64(N, H, W, bias):
    from upsnet_amd import ops
    torch.manual_seed(H + W)
    x = torch.randn(N, 3, H, W, device='cuda') * 50.0
    w = torch.randn(64, 3, 7, 7, device='cuda') * 0.02
    b = torch.randn(64, device='cuda') if bias else None
    out = ops.stem_pool_bf16(ops.image_to_nhwc4(x), ops.pack_stem_pool_weight_bf16(w), b)
    conv = F.relu(F.conv2d(x.bfloat16().double(), w.bfloat16().double(), None if b is None else b.double(), stride=2, padding=3))
    ref = F.max_pool2d(conv.to(torch.bfloat16).double(), 3, stride=2, padding=1)
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape and out.permute(0, 2, 3, 1).is_contiguous()
    # the kernel accumulates in fp32: a convolution value next to a bf16 rounding boundary may land on the other side (one step,
    # 2^-8 relative); everything else is exact
    err = (out.double() - ref).abs()
    assert float((err > 0).double().mean()) < 0.02
    assert bool((err <= ref.abs() * 2.0 ** -7 + 1e-6).all())


def test_backbone_stem_takes_the_fused_kernel_in_bf16_mode():
    """models/resnet.py: conv1 module in the bf16 mode = one stem_pool launch returning bf16; against the fp32 stem + library pool."""
    from upsnet_amd.models import hipconv
    from upsnet_amd.models.resnet import conv1, fold_frozen_bn
    torch.manual_seed(2)
    stem = conv1().cuda().eval()
    with torch.no_grad():
        stem.bn1.running_var.fill_(0.7)
        stem.bn1.running_mean.normal_(0, 0.1)
    fold_frozen_bn(stem)
    x = torch.randn(1, 3, 96, 160, device='cuda') * 40.0
    saved = hipconv.PRECISION
    try:
        with torch.no_grad():
            ref = stem(x)
            hipconv.PRECISION, hipconv.TRACE = 'bf16', []
            out = stem(x)
            trace, hipconv.TRACE = hipconv.TRACE, None
    finally:
        hipconv.PRECISION, hipconv.TRACE = saved, None
    assert [r['form'] for r in trace] == ['stem + pool bf16']
    assert ref.dtype == torch.float32 and out.dtype == torch.bfloat16 and out.shape == ref.shape
    assert float((out.float() - ref).abs().max()) <= 0.02 * float(ref.abs().max())


@pytest.mark.parametrize("N,H,W,bias", [(1, 64, 128, True), (2, 37, 53, True), (1, 200, 333, False), (1, 9, 7, True), (1, 130, 66, True), (1, 256, 512, True)])
def test_stem_pool_f32_vs_float64(N, H, W, bias):
    """fp32 form: exact fp32 products, fp32 sums -- 1e-4 against float64; odd sizes exercise partial tiles and the zero padding of both
    the convolution (pad 3) and the pool (pad 1, -inf semantics)."""
    from upsnet_amd import ops
    torch.manual_seed(H + W)
    x = torch.randn(N, 3, H, W, device='cuda') * 50.0
    w = torch.randn(64, 3, 7, 7, device='cuda') * 0.02
    b = torch.randn(64, device='cuda') if bias else None
    out = ops.stem_pool_f32(ops.image_to_nhwc4(x), ops.pack_stem_pool_weight_f32(w), b)
    ref = F.max_pool2d(F.relu(F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=2, padding=3)), 3, stride=2, padding=1)
    assert out.dtype == torch.float32 and out.shape == ref.shape and out.permute(0, 2, 3, 1).is_contiguous()
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_backbone_stem_takes_the_fused_kernel_in_fp32_mode():
    """models/resnet.py: the conv1 module = ONE stem_pool launch (no library max-pool); hipconv.STEM_POOL = False restores the stem kernel +
    max_pool2d; both within 1e-4 of each other's float64 value (different summation orders)."""
    from upsnet_amd.models import hipconv
    from upsnet_amd.models.resnet import conv1, fold_frozen_bn
    torch.manual_seed(4)
    stem = conv1().cuda().eval()
    with torch.no_grad():
        stem.bn1.running_var.fill_(0.7)
        stem.bn1.running_mean.normal_(0, 0.1)
    fold_frozen_bn(stem)
    x = torch.randn(1, 3, 96, 160, device='cuda') * 40.0
    saved = hipconv.STEM_POOL
    try:
        with torch.no_grad():
            hipconv.TRACE = []
            out = stem(x)
            trace, hipconv.TRACE = hipconv.TRACE, None
            hipconv.STEM_POOL = False
            sep = stem(x)
    finally:
        hipconv.STEM_POOL, hipconv.TRACE = saved, None
    assert [r['form'] for r in trace] == ['stem + pool']
    assert out.dtype == torch.float32 and out.shape == sep.shape
    np.testing.assert_allclose(out.cpu().numpy(), sep.cpu().numpy(), rtol=1e-4, atol=1e-4)
