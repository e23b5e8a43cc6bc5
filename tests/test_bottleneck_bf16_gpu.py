"""GPU: the one-launch identity bottleneck of the bf16 mode (csrc/bottleneck_bf16.hip) vs (a) a float64 restatement of the block
(upsnet/models/resnet.py:84-100) that rounds to bf16 where the kernel rounds, and (b) the three-launch bf16 path it replaces."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


def _block_weights(cm, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    c = 4 * cm
    w1 = torch.randn(cm, c, 1, 1, generator=g) / c ** 0.5
    w2 = torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5
    w3 = torch.randn(c, cm, 1, 1, generator=g) / cm ** 0.5
    b1, b2, b3 = torch.randn(cm, generator=g) * 0.1, torch.randn(cm, generator=g) * 0.1, torch.randn(c, generator=g) * 0.1
    return [t.cuda() for t in (w1, w2, w3, b1, b2, b3)]


def _reference(x16, w1, w2, w3, b1, b2, b3):
    """float64 products of the bf16-rounded operands, bias / shortcut / ReLU in float64, rounded to bf16 after every layer."""
    d = lambda t: _bf(t).double()
    t1 = _bf(F.relu(F.conv2d(x16.double(), d(w1), b1.double())))
    t2 = _bf(F.relu(F.conv2d(t1.double(), d(w2), b2.double(), padding=1)))
    return F.relu(F.conv2d(t2.double(), d(w3), b3.double()) + x16.double())


@pytest.mark.parametrize("cm,N,H,W", [
    (64, 1, 32, 48),       # whole tiles
    (64, 2, 37, 53),       # ragged tiles in both directions, two images
    (128, 1, 24, 40),
    (128, 1, 19, 21),
    (256, 2, 13, 22),
    (256, 1, 3, 5),        # map smaller than one tile
    (512, 1, 8, 16),
    (512, 2, 9, 7),
])
def test_bottleneck_bf16_vs_float64(cm, N, H, W):
    from upsnet_amd import ops
    torch.manual_seed(cm + H)
    ws = _block_weights(cm, cm + W)
    x = _bf(torch.randn(N, 4 * cm, H, W, device='cuda')).contiguous(memory_format=torch.channels_last)
    out = ops.bottleneck_bf16(x, ops.pack_bottleneck_bf16(*ws))
    assert out.dtype == torch.bfloat16 and out.shape == x.shape
    ref = _reference(x, *ws)
    # the kernel accumulates in fp32 and rounds t1 / t2 / out to bf16: an intermediate that lands on the other side of a rounding
    # boundary moves the result by one bf16 step of a term of the next layer's sum -- small against the output's scale
    err = (out.double() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) <= 0.02 * scale, (float(err.max()), scale)
    assert float(err.mean()) <= 2e-3 * scale
    # the rounding of the last layer alone bounds most elements: within one bf16 step (2^-8 relative) of the reference
    near = err <= ref.abs() * 2.0 ** -7 + 1e-2 * scale * 2.0 ** -4
    assert float(near.double().mean()) > 0.99


@pytest.mark.parametrize("cm,N,H,W", [(64, 1, 40, 72), (128, 2, 21, 35), (256, 1, 16, 24), (512, 1, 9, 12)])
def test_bottleneck_bf16_equals_the_three_launch_path(cm, N, H, W):
    """Same rounding points and the same K order as upsnet_conv2d_nhwc_bf16 run three times with bf16 activations: same bits."""
    from upsnet_amd import ops
    torch.manual_seed(cm)
    w1, w2, w3, b1, b2, b3 = _block_weights(cm, 7 * cm)
    c = 4 * cm
    x = _bf(torch.randn(N, c, H, W, device='cuda')).contiguous(memory_format=torch.channels_last)
    fused = ops.bottleneck_bf16(x, ops.pack_bottleneck_bf16(w1, w2, w3, b1, b2, b3))
    p1, p2, p3 = (ops.pack_conv_weight_bf16(w, split=False) for w in (w1, w2, w3))
    bf = torch.bfloat16
    t1 = ops.conv2d_nhwc_bf16_multi([x], p1[0], None, p1[2], b1, cm, 1, 1, 0, relu=True, out_dtype=bf)[0]
    t2 = ops.conv2d_nhwc_bf16_multi([t1], p2[0], None, p2[2], b2, cm, 3, 1, 1, relu=True, out_dtype=bf)[0]
    y = ops.conv2d_nhwc_bf16_multi([t2], p3[0], None, p3[2], b3, c, 1, 1, 0, relu=True, residuals=[x], out_dtype=bf)[0]
    diff = (fused.float() - y.float()).abs()
    mism = float((diff > 0).float().mean())
    # (a split-K instance of the separate kernels sums its partial accumulators in another order: allow isolated one-step flips)
    assert mism < 1e-3, mism
    assert float(diff.max()) <= 2.0 ** -6 * float(y.float().abs().max())


def _proj_weights(cm, cin, seed):
    g = torch.Generator(device='cpu').manual_seed(seed)
    c = 4 * cm
    w1 = torch.randn(cm, cin, 1, 1, generator=g) / cin ** 0.5
    w2 = torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5
    w3 = torch.randn(c, cm, 1, 1, generator=g) / cm ** 0.5
    wd = torch.randn(c, cin, 1, 1, generator=g) / cin ** 0.5
    bs = [torch.randn(n, generator=g) * 0.1 for n in (cm, cm, c, c)]
    return [t.cuda() for t in (w1, w2, w3, wd)] + [t.cuda() for t in bs]


@pytest.mark.parametrize("cm,cin,stride,N,H,W", [
    (64, 64, 1, 1, 32, 48),        # res2: stride 1
    (64, 64, 1, 2, 19, 37),        # ragged tiles, two images
    (128, 256, 2, 1, 48, 64),      # res3: stride 2
    (128, 256, 2, 1, 37, 53),      # odd input size: H = (Hin - 1) / 2 + 1
    (256, 512, 2, 1, 24, 40),      # res4
    (256, 512, 2, 2, 9, 15),
])
def test_bottleneck_proj_bf16_vs_float64(cm, cin, stride, N, H, W):
    """First bottleneck of a stage (projection shortcut, resnet.py:84-100) in one launch: conv3 and the projection are one GEMM over
    [t2 ; x], so the shortcut is accumulated in fp32. Reference: float64 on the bf16-rounded operands, t1 / t2 rounded to bf16."""
    from upsnet_amd import ops
    torch.manual_seed(cm + H)
    w1, w2, w3, wd, b1, b2, b3, bd = _proj_weights(cm, cin, cm + W)
    x = _bf(torch.randn(N, cin, H, W, device='cuda')).contiguous(memory_format=torch.channels_last)
    out = ops.bottleneck_proj_bf16(x, ops.pack_bottleneck_proj_bf16(w1, w2, w3, wd, b1, b2, b3, bd), stride)
    d = lambda t: _bf(t).double()
    t1 = _bf(F.relu(F.conv2d(x.double(), d(w1), b1.double(), stride=stride)))
    t2 = _bf(F.relu(F.conv2d(t1.double(), d(w2), b2.double(), padding=1)))
    ref = F.relu(F.conv2d(t2.double(), d(w3), b3.double()) + F.conv2d(x.double(), d(wd), bd.double(), stride=stride))
    assert out.dtype == torch.bfloat16 and out.shape == ref.shape and out.permute(0, 2, 3, 1).is_contiguous()
    err = (out.double() - ref).abs()
    scale = float(ref.abs().max())
    assert float(err.max()) <= 0.02 * scale, (float(err.max()), scale)
    assert float(err.mean()) <= 2e-3 * scale
    near = err <= ref.abs() * 2.0 ** -7 + 1e-2 * scale * 2.0 ** -4
    assert float(near.double().mean()) > 0.99


def test_backbone_stage_takes_the_block_kernels_in_bf16_mode():
    """models/resnet.py: in the bf16 mode every bottleneck of a stage runs as ONE launch (hipconv.block: the projection block and the
    identity blocks); hipconv.BF16_BLOCK = False restores the separate layers, which round the shortcut of the projection block to bf16
    before the add (the fused block accumulates it in fp32): the two agree within bf16 steps of the output."""
    from upsnet_amd.models import hipconv
    from upsnet_amd.models.resnet import res_block, fold_frozen_bn
    torch.manual_seed(5)
    stage = res_block(128, 3, stride=2).cuda().eval()
    with torch.no_grad():
        for m in stage.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_var.fill_(0.5)
                m.weight.fill_(0.7)
    fold_frozen_bn(stage)
    stage = stage.to(memory_format=torch.channels_last)
    x = torch.randn(1, 256, 48, 64, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    saved = (hipconv.PRECISION, hipconv.BF16_MIN_WG, hipconv.BF16_BLOCK, hipconv.BF16_BLOCK_MIN_TILES, hipconv.BF16_PROJ)
    try:
        hipconv.PRECISION, hipconv.BF16_MIN_WG, hipconv.BF16_BLOCK_MIN_TILES = 'bf16', 0, 0   # (a test-size map has few tiles)
        with torch.no_grad():
            hipconv.BF16_BLOCK, hipconv.TRACE = True, []
            out = stage(x)
            trace, hipconv.TRACE = hipconv.TRACE, None
            hipconv.BF16_PROJ, hipconv.TRACE = False, []
            mid = stage(x)
            trace_mid, hipconv.TRACE = hipconv.TRACE, None
            hipconv.BF16_BLOCK = False
            sep = stage(x)
    finally:
        hipconv.PRECISION, hipconv.BF16_MIN_WG, hipconv.BF16_BLOCK, hipconv.BF16_BLOCK_MIN_TILES, hipconv.BF16_PROJ = saved
        hipconv.TRACE = None
    assert [r['form'] for r in trace] == ['bottleneck_proj_bf16', 'bottleneck_bf16', 'bottleneck_bf16']
    kinds = [r['form'] for r in trace_mid]
    assert kinds.count('bottleneck_bf16') == 2 and len(kinds) == 4 + 2, kinds
    assert out.dtype == torch.bfloat16 and sep.dtype == torch.bfloat16 and mid.dtype == torch.bfloat16
    top = float(sep.float().abs().max())
    for a in (out, mid):
        diff = (a.float() - sep.float()).abs()
        assert float(diff.max()) <= 2.0 ** -5 * top
        assert float(diff.mean()) <= 2.0 ** -9 * top
