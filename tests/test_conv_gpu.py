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


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride,relu,res,bias,bn", [
    (1, 256, 64, 40, 56, 1, True, None, True, 0),        # res2 conv1 shape family (BN = 64: Cout <= 64)
    (1, 64, 256, 40, 56, 1, True, 'same', True, 0),      # conv3 + shortcut
    (1, 256, 512, 41, 57, 2, False, None, True, 0),      # strided downsample, odd size, tail tile
    (1, 512, 128, 33, 31, 1, True, None, False, 128),
    (1, 512, 128, 33, 31, 1, True, None, False, 64),
    (2, 1024, 256, 16, 24, 1, False, 'up', True, 0),     # FPN lateral + top-down add through the nearest x2 upsampling
    (1, 2048, 256, 16, 16, 1, False, 'up', True, 128),
    (1, 32, 96, 7, 5, 1, True, 'same', True, 0),         # one K step, Cout not a multiple of 64, map smaller than a tile
    (3, 96, 160, 9, 9, 2, True, None, True, 128),
    (1, 160, 64, 21, 20, 1, True, 'same', True, 0),      # odd slab count
])
def test_conv1x1_gemm_kernel_vs_torch(N, Cin, Cout, H, W, stride, relu, res, bias, bn):
    """csrc/conv1x1.hip (the lean fp32 MFMA GEMM the 1x1 layers of the backbone / FPN run on) vs torch float64, 1e-4; its two tile
    forms agree bit for bit (same K order)."""
    from upsnet_amd import ops
    from upsnet_amd._lib import lib
    torch.manual_seed(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride)
    r = None
    if res == 'same':
        r = torch.randn_like(ref).float()
        ref = ref + r.double()
    elif res == 'up':
        r = torch.randn(N, Cout, ref.shape[2] // 2, ref.shape[3] // 2, device='cuda')
        ref = ref + F.interpolate(r.double(), scale_factor=2, mode='nearest')
    if relu:
        ref = ref.clamp_min(0)
    wp = ops.pack_conv1x1_weight(w)
    outs = []
    try:
        for t in ((bn,) if bn else (64, 128)):
            lib().upsnet_conv1x1_tuning(t)
            outs.append(ops.conv1x1_frag(x, wp, b, Cout, stride, relu=relu, residual=r, residual_up=(res == 'up')))
    finally:
        lib().upsnet_conv1x1_tuning(0)
    assert outs[0].shape == ref.shape
    np.testing.assert_allclose(outs[0].cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    for o in outs[1:]:
        assert torch.equal(o, outs[0])
    out2 = ops.conv1x1_frag(x.contiguous(memory_format=torch.channels_last), wp, b, Cout, stride, relu=relu, residual=r, residual_up=(res == 'up'))
    assert torch.equal(out2, outs[-1] if not bn else out2)


@pytest.mark.parametrize("N,H,W,C1", [(1, 64, 96, 512), (2, 37, 41, 512), (1, 9, 7, 128), (1, 128, 256, 512)])
def test_conv1x1_pair_kernel_res3(N, H, W, C1):
    """r10: the (C0, C2) = (128, 128) instance of csrc/conv1x1_pair.hip (res3: 128 -> 512 -> 128): bit-identical to two launches of
    csrc/conv1x1.hip, 1e-4 vs float64; the last case is the stage's real map at 1024x2048."""
    from upsnet_amd import ops
    torch.manual_seed(N + H + W + C1)
    x = torch.randn(N, 128, H, W, device='cuda').relu()
    sc = torch.randn(N, C1, H, W, device='cuda')
    w3 = torch.randn(C1, 128, 1, 1, device='cuda') / 128 ** 0.5
    b3 = torch.randn(C1, device='cuda')
    w1 = torch.randn(128, C1, 1, 1, device='cuda') / C1 ** 0.5
    b1 = torch.randn(128, device='cuda')
    p3, p1 = ops.pack_conv1x1_weight(w3), ops.pack_conv1x1_weight(w1)
    o1, o2 = ops.conv1x1_pair(x, sc, p3, b3, C1, p1, b1, 128)
    s1 = ops.conv1x1_frag(x, p3, b3, C1, 1, relu=True, residual=sc)
    s2 = ops.conv1x1_frag(s1, p1, b1, 128, 1, relu=True)
    assert torch.equal(o1, s1) and torch.equal(o2, s2)
    r1 = (F.conv2d(x.double(), w3.double(), b3.double()) + sc.double()).clamp_min(0)
    r2 = F.conv2d(r1, w1.double(), b1.double()).clamp_min(0)
    np.testing.assert_allclose(o1.cpu().numpy(), r1.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o2.cpu().numpy(), r2.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,H,W,C1", [(1, 32, 48, 1024), (2, 19, 21, 1024), (1, 5, 3, 256), (1, 11, 13, 384), (1, 64, 128, 1024)])
@pytest.mark.parametrize("waves", [8, 4])
def test_conv1x1_pair_kernel_res4(N, H, W, C1, waves, monkeypatch):
    """r10: the (C0, C2) = (256, 256) instance on 32-pixel tiles (conv1x1_pair32_f32_kernel, res4: 256 -> 1024 -> 256): bit-identical to two
    launches of csrc/conv1x1.hip, 1e-4 vs float64; ragged tiles, a map smaller than a tile, and the stage's real map at 1024x2048. Both
    workgroup forms: 8 waves (256-channel chunks; falls back to 4 when C1 % 256 != 0) and 4 waves."""
    from upsnet_amd import ops
    monkeypatch.setattr(ops, 'PAIR32_WAVES', waves)
    torch.manual_seed(N + H + W + C1)
    x = torch.randn(N, 256, H, W, device='cuda').relu()
    sc = torch.randn(N, C1, H, W, device='cuda')
    w3 = torch.randn(C1, 256, 1, 1, device='cuda') / 16
    b3 = torch.randn(C1, device='cuda')
    w1 = torch.randn(256, C1, 1, 1, device='cuda') / C1 ** 0.5
    b1 = torch.randn(256, device='cuda')
    p3, p1 = ops.pack_conv1x1_weight(w3), ops.pack_conv1x1_weight(w1)
    o1, o2 = ops.conv1x1_pair(x, sc, p3, b3, C1, p1, b1, 256)
    s1 = ops.conv1x1_frag(x, p3, b3, C1, 1, relu=True, residual=sc)
    s2 = ops.conv1x1_frag(s1, p1, b1, 256, 1, relu=True)
    assert torch.equal(o1, s1) and torch.equal(o2, s2)
    r1 = (F.conv2d(x.double(), w3.double(), b3.double()) + sc.double()).clamp_min(0)
    r2 = F.conv2d(r1, w1.double(), b1.double()).clamp_min(0)
    np.testing.assert_allclose(o1.cpu().numpy(), r1.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o2.cpu().numpy(), r2.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,Cin,H,W,ca,cb,stride", [(1, 64, 40, 56, 64, 256, 1), (2, 256, 37, 41, 128, 512, 2), (1, 1024, 16, 32, 512, 2048, 2),
                                                     (1, 512, 9, 7, 256, 1024, 2), (1, 32, 33, 17, 96, 40, 1), (1, 64, 256, 512, 64, 256, 1)])
def test_conv1x1_sibling_launch(N, Cin, H, W, ca, cb, stride):
    """csrc/conv1x1.hip, sibling mode: conv1 (+ ReLU) and the projection shortcut of a stage's first bottleneck (same input, same stride)
    in one launch over the concatenated output channels -- each output bit-identical to a launch of its own, 1e-4 vs float64; both tile
    widths, a split inside a 128-channel tile (64 + 256), a ragged last column block, the res2 map at 1024x2048."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + ca)
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wa, wb = torch.randn(ca, Cin, 1, 1, device='cuda') / Cin ** 0.5, torch.randn(cb, Cin, 1, 1, device='cuda') / Cin ** 0.5
    ba, bb = torch.randn(ca, device='cuda'), torch.randn(cb, device='cuda')
    ya, yb = ops.conv1x1_siblings(x, ops.pack_conv1x1_weight(torch.cat([wa, wb])), torch.cat([ba, bb]), ca, cb, stride, relu_a=True, relu_b=False)
    assert ops.last_kernel_form() == 'conv1x1_siblings<%d+%d>' % (ca, cb)
    sa = ops.conv1x1_frag(x, ops.pack_conv1x1_weight(wa), ba, ca, stride, relu=True)
    sb = ops.conv1x1_frag(x, ops.pack_conv1x1_weight(wb), bb, cb, stride, relu=False)
    assert torch.equal(ya, sa) and torch.equal(yb, sb)
    ra = F.conv2d(x.double(), wa.double(), ba.double(), stride=stride).clamp_min(0)
    rb = F.conv2d(x.double(), wb.double(), bb.double(), stride=stride)
    np.testing.assert_allclose(ya.cpu().numpy(), ra.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(yb.cpu().numpy(), rb.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    y0, _ = ops.conv1x1_siblings(x, ops.pack_conv1x1_weight(torch.cat([wa, wb])), None, ca, cb, stride, relu_a=False, relu_b=True)
    assert torch.equal(y0, ops.conv1x1_frag(x, ops.pack_conv1x1_weight(wa), None, ca, stride, relu=False))


@pytest.mark.parametrize("N,H,W,C1", [(1, 64, 96, 256), (2, 37, 41, 256), (1, 9, 7, 128), (1, 40, 40, 384)])
def test_conv1x1_pair_kernel(N, H, W, C1):
    """csrc/conv1x1_pair.hip: relu(conv3(x) + b3 + shortcut) and relu(conv1(that) + b1) in one launch -- bit-identical to two launches
    of csrc/conv1x1.hip (same K order), within 1e-4 of torch float64; tail tiles and maps smaller than a tile included."""
    from upsnet_amd import ops
    torch.manual_seed(N + H + W + C1)
    x = torch.randn(N, 64, H, W, device='cuda').relu()
    sc = torch.randn(N, C1, H, W, device='cuda')
    w3 = torch.randn(C1, 64, 1, 1, device='cuda') / 8
    b3 = torch.randn(C1, device='cuda')
    w1 = torch.randn(64, C1, 1, 1, device='cuda') / C1 ** 0.5
    b1 = torch.randn(64, device='cuda')
    p3, p1 = ops.pack_conv1x1_weight(w3), ops.pack_conv1x1_weight(w1)
    o1, o2 = ops.conv1x1_pair(x, sc, p3, b3, C1, p1, b1, 64)
    s1 = ops.conv1x1_frag(x, p3, b3, C1, 1, relu=True, residual=sc)
    s2 = ops.conv1x1_frag(s1, p1, b1, 64, 1, relu=True)
    assert torch.equal(o1, s1) and torch.equal(o2, s2)
    r1 = (F.conv2d(x.double(), w3.double(), b3.double()) + sc.double()).clamp_min(0)
    r2 = F.conv2d(r1, w1.double(), b1.double()).clamp_min(0)
    np.testing.assert_allclose(o1.cpu().numpy(), r1.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(o2.cpu().numpy(), r2.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        ops.conv1x1_pair(x, sc[:, :64], p3, b3, C1, p1, b1, 64)


def test_res2_stage_with_paired_boundaries_is_bit_identical():
    """models/resnet.py res_block: with the block-boundary pairs (hipconv.use_pair) the res2 stage returns the very same bits as
    with one launch per layer, and takes two launches fewer."""
    from upsnet_amd import ops
    from upsnet_amd.models import hipconv, resnet
    torch.manual_seed(3)
    stage = resnet.res_block(64, 3).cuda().eval()
    for m in stage.modules():
        if isinstance(m, torch.nn.Conv2d):
            torch.nn.init.normal_(m.weight, std=(2.0 / (m.in_channels * m.kernel_size[0] ** 2)) ** 0.5)
    resnet.fold_frozen_bn(stage)
    stage = stage.to(memory_format=torch.channels_last)
    x = torch.randn(1, 64, 256, 512, device='cuda').relu().contiguous(memory_format=torch.channels_last)
    outs, launches = [], []
    old = hipconv.PAIR
    try:
        for pair in (True, False):
            hipconv.PAIR = pair
            ops.PROFILE['events'], ops.PROFILE['enabled'] = [], True
            with torch.no_grad():
                outs.append(stage(x))
            launches.append([e[5] if len(e) > 5 else '' for e in ops.PROFILE['events']])
    finally:
        hipconv.PAIR = old
        ops.PROFILE['enabled'] = False
        ops.PROFILE['events'] = []
    assert torch.equal(outs[0], outs[1])
    assert len(launches[0]) == len(launches[1]) - 2 and sum('pair' in l for l in launches[0]) == 2


def test_hipconv_routes_1x1_layers_to_the_gemm_kernel():
    from upsnet_amd import ops
    from upsnet_amd.models import hipconv
    torch.manual_seed(0)
    m = torch.nn.Conv2d(256, 512, 1, stride=2, bias=True).cuda().to(memory_format=torch.channels_last)
    x = torch.randn(1, 256, 128, 256, device='cuda').contiguous(memory_format=torch.channels_last)
    small = torch.randn(1, 256, 8, 8, device='cuda').contiguous(memory_format=torch.channels_last)
    assert hipconv._use_conv1x1(m, x) and not hipconv._use_conv1x1(m, small)
    ops.PROFILE['events'], ops.PROFILE['enabled'] = [], True
    try:
        with torch.no_grad():
            y = hipconv.conv(m, x, relu=True)
            ys = hipconv.conv(m, small, relu=True)
    finally:
        ops.PROFILE['enabled'] = False
    kinds = [e[5] for e in ops.PROFILE['events']]
    assert '(gemm)' in kinds[0] and '(gemm)' not in kinds[1], kinds
    with torch.no_grad():
        np.testing.assert_allclose(y.cpu().numpy(), F.relu(m(x)).cpu().numpy(), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(ys.cpu().numpy(), F.relu(m(small)).cpu().numpy(), rtol=1e-4, atol=1e-4)


def test_conv_multi_map_launch_with_per_map_weights():
    """upsnet_conv2d_nhwc_f32_multiw (the four per-level score products of the FCN head in one launch): bit-identical to one launch per map."""
    from upsnet_amd import ops
    torch.manual_seed(7)
    xs = [torch.randn(1, 128, 64 >> l, 96 >> l, device='cuda') for l in range(4)]
    ws = [torch.randn(19, 128, 1, 1, device='cuda') / 128 ** 0.5 for _ in range(4)]
    packs = [ops.pack_conv_weight(w) for w in ws]
    one = [ops.conv2d_nhwc(x, wp, ldw, None, 19, 1, 1, 0) for x, (wp, ldw) in zip(xs, packs)]
    multi = ops.conv2d_nhwc_multiw(xs, [wp for wp, _ in packs], packs[0][1], 19, 1, 1, 0)
    for a, b, x, w in zip(one, multi, xs, ws):
        assert torch.equal(a, b)
        np.testing.assert_allclose(b.cpu().numpy(), F.conv2d(x.double(), w.double()).float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("S,H,W,nlev,bias", [(19, 32, 64, 4, True), (133, 24, 40, 4, True), (5, 8, 8, 2, False), (19, 16, 16, 1, True)])
def test_fcn_score_combine_vs_oracle(S, H, W, nlev, bias):
    """Bit-exact vs the C oracle (same expression order, no FMA)."""
    import oracle
    from upsnet_amd import ops
    rng = np.random.default_rng(S + H)
    parts = [rng.standard_normal((H >> l, W >> l, S)).astype(np.float32) for l in range(nlev)]
    b = rng.standard_normal(S).astype(np.float32) if bias else None
    ref = oracle.fcn_score_combine(parts, b)
    tp = [torch.from_numpy(t).cuda().permute(2, 0, 1)[None] for t in parts]   # logical NCHW over NHWC memory
    out = ops.fcn_score_combine(tp, None if b is None else torch.from_numpy(b).cuda())
    assert out.shape == (1, S, H, W)
    np.testing.assert_array_equal(out[0].permute(1, 2, 0).cpu().numpy(), ref)


def test_fcn_head_commuted_score_vs_reference_order():
    """FCNHead.forward_score with the 1x1 conv commuted below the upsampling == the reference op order (fcn.py:94-100), 1e-4."""
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    from upsnet_amd.models.fcn import FCNHead
    torch.manual_seed(3)
    head = FCNHead(256, 19, 2).cuda().eval()
    torch.nn.init.normal_(head.score.weight, 0, 0.05)
    torch.nn.init.normal_(head.score.bias, 0, 0.5)
    for layer in head.fcn_subnet.conv:
        torch.nn.init.normal_(layer[0].conv_offset.weight, 0, 0.01)
    feats = [torch.randn(1, 256, 64 >> l, 96 >> l, device='cuda') for l in range(4)]
    with torch.no_grad():
        a = head.forward_score(*feats, commute=True)
        b = head.forward_score(*feats, commute=False)
        # and against plain torch ops on the subnet outputs (the reference's own sequence)
        ys = head.fcn_subnet.forward_levels(feats)
        ups = [ys[0]] + [F.interpolate(ys[l], None, 2 ** l, mode='bilinear', align_corners=False) for l in (1, 2, 3)]
        c = F.conv2d(torch.cat([u.contiguous() for u in ups], 1), head.score.weight, head.score.bias)
    assert a.shape == b.shape == c.shape == (1, 19, 64, 96)
    np.testing.assert_allclose(a.cpu().numpy(), c.cpu().numpy(), rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(b.cpu().numpy(), c.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(1, 64, 256, 16, 24), (2, 32, 64, 6, 10), (1, 256, 256, 64, 128)])
def test_conv_residual_nearest_upsample_fused(N, Cin, Cout, H, W):
    """FPN top-down add: conv1x1(x) + nearest_up2(residual) with the upsampling folded into the residual read (fpn.py:34,90-96)."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda')
    r = torch.randn(N, Cout, H // 2, W // 2, device='cuda')
    wp, ldw = ops.pack_conv_weight(w)
    out = ops.conv2d_nhwc(x, wp, ldw, b, Cout, 1, 1, 0, residual=r, residual_up=True)
    mat = ops.conv2d_nhwc(x, wp, ldw, b, Cout, 1, 1, 0, residual=F.interpolate(r, scale_factor=2, mode='nearest'))
    assert torch.equal(out, mat)   # same kernel, same operands: identical bits
    ref = F.conv2d(x.double(), w.double(), b.double()) + F.interpolate(r, scale_factor=2, mode='nearest').double()
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc(x, wp, ldw, b, Cout, 1, 1, 0, residual=r)   # shape mismatch without residual_up


@pytest.mark.parametrize("shapes,Cin,Cout,relu,res,bias", [
    ([(1, 32, 64)], 64, 64, True, False, True),
    ([(1, 33, 47)], 32, 96, False, False, True),          # odd sizes: partial 2x2 tiles on the right / bottom edge
    ([(2, 14, 14)], 256, 256, True, True, True),          # batch > 1, residual, tiles wrap rows every 7
    ([(1, 17, 9)], 64, 18, False, False, False),          # Cout <= 32: the 32-tile x 32-channel form (4 waves, 4 positions each)
    ([(1, 64, 128), (1, 32, 64), (1, 16, 32)], 256, 18, False, False, True),   # DCN offset conv over 3 levels
    ([(2, 9, 11)], 32, 32, True, True, True),             # exactly 32 channels, residual, ReLU
    ([(1, 64, 128), (1, 32, 64), (1, 16, 32), (1, 8, 16), (1, 4, 8)], 256, 256, True, False, True),   # 5 maps, one launch (RPN head)
])
def test_winograd_conv_vs_torch(shapes, Cin, Cout, relu, res, bias):
    """Fused Winograd F(2x2,3x3) vs torch fp64 conv2d (1e-4) and vs the direct MFMA kernel (same tolerance, much closer in practice)."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + Cout + len(shapes))
    xs = [torch.randn(n, Cin, h, w, device='cuda') for n, h, w in shapes]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    rs = [torch.randn(x.shape[0], Cout, x.shape[2], x.shape[3], device='cuda') for x in xs] if res else None
    wp, ldw = ops.pack_winograd_weight(w)
    outs = ops.conv2d_winograd_multi(xs, wp, ldw, b, Cout, relu=relu, residuals=rs)
    wd, ldd = ops.pack_conv_weight(w)
    direct = ops.conv2d_nhwc_multi(xs, wd, ldd, b, Cout, 3, 1, 1, relu=relu, residuals=rs)
    for i, (x, o, d) in enumerate(zip(xs, outs, direct)):
        ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
        if res:
            ref = ref + rs[i].double()
        if relu:
            ref = ref.clamp_min(0)
        assert o.shape == ref.shape
        np.testing.assert_allclose(o.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
        assert float((o - d).abs().max()) < 2e-5
    again = ops.conv2d_winograd_multi(xs, wp, ldw, b, Cout, relu=relu, residuals=rs)
    assert all(torch.equal(a, o) for a, o in zip(again, outs))   # fixed summation order: bit-repeatable


@pytest.mark.parametrize("shapes,Cin,Cout", [
    ([(1, 1, 1)], 32, 5),                       # a single partial tile, odd Cout
    ([(1, 2, 3), (3, 5, 7)], 64, 70),           # ragged maps, batch 3, Cout just above one 64-channel tile
    ([(2, 31, 17)], 96, 64),                    # Cin = 6 slabs of 16
])
def test_winograd_tile_forms_agree_bit_for_bit(shapes, Cin, Cout):
    """The 64-tile (8 waves, one workgroup per CU) and the 32-tile (4 waves, two per CU) forms of the Winograd kernel compute
    every output element with the same arithmetic in the same order: identical bits, whatever the launcher would pick; both
    within 1e-4 of torch fp64 on edge shapes."""
    from upsnet_amd import ops
    from upsnet_amd._lib import lib
    torch.manual_seed(Cin + Cout)
    xs = [torch.randn(n, Cin, h, w, device='cuda') for n, h, w in shapes]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device='cuda')
    rs = [torch.randn(x.shape[0], Cout, x.shape[2], x.shape[3], device='cuda') for x in xs]
    wp, ldw = ops.pack_winograd_weight(w)
    outs = {}
    try:
        for tm in (64, 32):
            lib().upsnet_conv_tuning(tm, 0)
            outs[tm] = ops.conv2d_winograd_multi(xs, wp, ldw, b, Cout, relu=True, residuals=rs)
    finally:
        lib().upsnet_conv_tuning(0, 0)
    for x, r, o64, o32 in zip(xs, rs, outs[64], outs[32]):
        assert torch.equal(o64, o32)
        ref = (F.conv2d(x.double(), w.double(), b.double(), padding=1) + r.double()).clamp_min(0)
        np.testing.assert_allclose(o64.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,H,W,Cin,Cout,ksplit,relu,res", [
    (1, 64, 128, 256, 256, 2, True, False),     # res4 3x3
    (1, 32, 64, 512, 512, 4, True, False),      # res5 3x3
    (1, 33, 47, 64, 96, 2, False, True),        # odd sizes, residual, Cout not a multiple of 64
    (2, 14, 14, 256, 256, 8, True, True),       # 16 slabs split 8 ways
    (1, 9, 7, 96, 32, 3, False, False),         # 6 slabs split 3 ways, 32-channel form
    (1, 20, 12, 128, 20, 2, True, True),        # 32-channel form, Cout % 4 == 0 but < 32
])
def test_winograd_splitk_vs_torch(N, H, W, Cin, Cout, ksplit, relu, res):
    """Winograd with the K walk split over `ksplit` workgroups per tile + the shared reduce kernel vs torch fp64 (1e-4) and
    vs the unsplit Winograd launch."""
    from upsnet_amd import ops
    torch.manual_seed(H + W + Cin + ksplit)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device='cuda')
    r = torch.randn(N, Cout, H, W, device='cuda') if res else None
    wp, ldw = ops.pack_winograd_weight(w)
    out = ops.conv2d_winograd_splitk(x, wp, ldw, b, Cout, ksplit, relu=relu, residual=r)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    one = ops.conv2d_winograd_multi([x], wp, ldw, b, Cout, relu=relu, residuals=None if r is None else [r])[0]
    assert float((out - one).abs().max()) < 2e-5
    assert torch.equal(out, ops.conv2d_winograd_splitk(x, wp, ldw, b, Cout, ksplit, relu=relu, residual=r))   # bit-repeatable
    with pytest.raises(RuntimeError):
        ops.conv2d_winograd_splitk(x, wp, ldw, b, Cout, 9, relu=relu, residual=r)


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride,pad,relu,res", [
    (1, 64, 64, 33, 47, 3, 1, 1, True, False),
    (1, 256, 64, 40, 56, 1, 1, 0, True, False),
    (1, 64, 256, 40, 56, 1, 1, 0, True, True),
    (1, 256, 128, 41, 57, 1, 2, 0, False, False),
    (2, 256, 256, 14, 14, 3, 1, 1, True, False),
    (1, 512, 19, 24, 40, 1, 1, 0, False, False),
    (1, 2048, 256, 8, 16, 1, 1, 0, False, False),
    (1, 128, 160, 37, 45, 3, 1, 1, True, True),       # haloed-patch 3x3 kernel: partial 8x16 tiles on both edges, Cout not % 128
    (3, 32, 128, 9, 17, 3, 1, 1, False, False),       # batch > 1, one K slab, map barely larger than a tile
    (1, 64, 64, 16, 32, 3, 2, 1, True, False),        # strided 3x3 stays on the general kernel
])
@pytest.mark.parametrize("split", [True, False])
def test_conv_bf16_matrix_cores_vs_torch(N, Cin, Cout, H, W, k, stride, pad, relu, res, split):
    """bf16 MFMA convolution: the 3-term split meets the fp32 tolerance (1e-4); plain bf16 (configs[2]) is checked at bf16 accuracy."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device='cuda')
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=pad)
    r = torch.randn_like(ref).float() if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    hi, lo, ldw = ops.pack_conv_weight_bf16(w, split=split)
    out = ops.conv2d_nhwc_bf16_multi([x], hi, lo, ldw, b, Cout, k, stride, pad, relu=relu, residuals=None if r is None else [r])[0]
    assert out.shape == ref.shape
    tol = 1e-4 if split else 3e-2
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=tol, atol=tol)
    again = ops.conv2d_nhwc_bf16_multi([x], hi, lo, ldw, b, Cout, k, stride, pad, relu=relu, residuals=None if r is None else [r])[0]
    assert torch.equal(again, out)


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride,pad,relu,res", [
    (1, 256, 64, 40, 56, 1, 1, 0, True, None), (1, 64, 256, 40, 56, 1, 1, 0, True, 'bf16'), (1, 256, 512, 41, 57, 1, 2, 0, False, None),
    (1, 512, 128, 24, 40, 1, 1, 0, True, 'fp32'), (2, 128, 128, 19, 33, 3, 1, 1, True, None), (1, 64, 64, 16, 32, 3, 1, 1, True, None),
    (1, 2048, 256, 8, 16, 1, 1, 0, False, None), (1, 128, 160, 37, 45, 3, 1, 1, False, 'bf16'),
])
@pytest.mark.parametrize("in16,out16", [(True, True), (True, False), (False, True)])
def test_conv_bf16_activations(N, Cin, Cout, H, W, k, stride, pad, relu, res, in16, out16):
    """bf16 mode with bf16 ACTIVATIONS between layers (r08): bf16 NHWC inputs are loaded unconverted (8 channels per 16 bytes), a
    bf16 residual is widened in the epilogue, and the fp32 accumulator + bias + residual + ReLU is rounded ONCE to the bf16 output.
    Reference: the same convolution in float64 on the bf16-rounded operands; a bf16 result must be within one rounding of it."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + Cout + H + k)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device='cuda')
    xin = x.bfloat16() if in16 else x
    Ho, Wo = (H + 2 * pad - k) // stride + 1, (W + 2 * pad - k) // stride + 1
    r = None
    if res is not None:
        r = torch.randn(N, Cout, Ho, Wo, device='cuda')
        r = r.bfloat16() if res == 'bf16' else r
    ref = F.conv2d(x.bfloat16().double(), w.bfloat16().double(), b.double(), stride=stride, padding=pad)
    if r is not None:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    hi, lo, ldw = ops.pack_conv_weight_bf16(w, split=False)
    out = ops.conv2d_nhwc_bf16_multi([xin], hi, None, ldw, b, Cout, k, stride, pad, relu=relu, residuals=None if r is None else [r],
                                     out_dtype=torch.bfloat16 if out16 else torch.float32)[0]
    assert out.shape == ref.shape and out.dtype == (torch.bfloat16 if out16 else torch.float32)
    assert out.permute(0, 2, 3, 1).is_contiguous()
    got, want = out.double().cpu().numpy(), ref.cpu().numpy()
    if not in16:      # fp32 input rounded inside the kernel: same operands as the reference
        pass
    tol = 2.0 ** -8 if out16 else 1e-4     # bf16: one rounding (8 significant bits) of an fp32-accurate value
    np.testing.assert_allclose(got, want, rtol=tol, atol=tol)
    again = ops.conv2d_nhwc_bf16_multi([xin], hi, None, ldw, b, Cout, k, stride, pad, relu=relu, residuals=None if r is None else [r],
                                       out_dtype=torch.bfloat16 if out16 else torch.float32)[0]
    assert torch.equal(again, out)


def test_backbone_block_keeps_bf16_activations_between_layers():
    """models/resnet.py in the bf16 mode: a bottleneck stage reads the fp32 stem output, runs every layer on the bf16 kernels with
    bf16 tensors in between (conv1 -> conv2 -> conv3 + bf16 shortcut) and returns bf16; against the fp32-mode stage within bf16
    accuracy of the activations' scale."""
    from upsnet_amd.models import hipconv
    from upsnet_amd.models.resnet import res_block, fold_frozen_bn
    torch.manual_seed(3)
    stage = res_block(128, 2, stride=2).cuda().eval()
    saved_min, hipconv.BF16_MIN_WG = hipconv.BF16_MIN_WG, 0        # (a test-size map has few tiles: take the bf16 kernels anyway)
    with torch.no_grad():
        for m in stage.modules():
            if isinstance(m, torch.nn.BatchNorm2d):
                m.running_var.fill_(0.5)
                m.weight.fill_(0.7)
    fold_frozen_bn(stage)
    stage = stage.to(memory_format=torch.channels_last)
    x = torch.randn(1, 256, 48, 64, device='cuda').contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        ref = stage(x)
        hipconv.PRECISION = 'bf16'
        try:
            assert hipconv.act_dtype() == torch.bfloat16
            hipconv.TRACE = []
            out = stage(x)
            trace, hipconv.TRACE = hipconv.TRACE, None
        finally:
            hipconv.PRECISION = 'fp32'
            hipconv.TRACE = None
            hipconv.BF16_MIN_WG = saved_min
    assert ref.dtype == torch.float32 and out.dtype == torch.bfloat16
    assert all(r['out'].dtype == torch.bfloat16 for r in trace) and trace[0]['x'].dtype == torch.float32 and trace[-1]['x'].dtype == torch.bfloat16
    scale = float(ref.abs().max())
    assert float((out.float() - ref).abs().max()) < 0.03 * scale


@pytest.mark.parametrize("N,Cin,Cout,H,W,split", [(1, 256, 256, 32, 64, True), (2, 512, 256, 18, 22, True), (1, 64, 96, 6, 10, False)])
def test_conv_bf16_lateral_with_upsampled_residual(N, Cin, Cout, H, W, split):
    """FPN top-down add on the bf16 kernel: conv1x1(x) + nearest_up2(residual) with the upsampling folded into the residual read
    (fpn.py:34,90-96); widths that are / are not a multiple of 32 (both index paths)."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + W)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda')
    r = torch.randn(N, Cout, H // 2, W // 2, device='cuda')
    ref = F.conv2d(x.double(), w.double(), b.double()) + F.interpolate(r.double(), scale_factor=2, mode='nearest')
    hi, lo, ldw = ops.pack_conv_weight_bf16(w, split=split)
    out = ops.conv2d_nhwc_bf16_multi([x], hi, lo, ldw, b, Cout, 1, 1, 0, residuals=[r], residual_up=True)[0]
    tol = 1e-4 if split else 3e-2
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=tol, atol=tol)
    with pytest.raises(RuntimeError):
        ops.conv2d_nhwc_bf16_multi([x], hi, lo, ldw, b, Cout, 1, 1, 0, residuals=[r[:, :, :-1]], residual_up=True)


@pytest.mark.parametrize("N,Cin,Cout,H,W,k,stride,ksplit,relu,res", [
    (1, 256, 256, 64, 128, 3, 1, 2, True, False), (1, 512, 512, 32, 64, 3, 1, 4, True, False), (1, 2048, 512, 32, 64, 1, 1, 4, True, False),
    (1, 1024, 256, 17, 23, 1, 1, 3, False, True), (2, 64, 128, 9, 9, 3, 2, 2, True, True),
    # narrow heads (ldw = 32, 128-pixel tiles, one-element reduce for Cout % 4 != 0): the offset predictors of UPSNet-101-DCN's res4 / res5
    # at 800x1333 (r10), a ragged small map, and a 4-divisible narrow head
    (1, 256, 18, 50, 84, 3, 1, 8, False, False), (1, 512, 18, 25, 42, 3, 1, 8, False, False), (1, 128, 18, 13, 11, 3, 1, 4, False, True),
    (2, 64, 12, 9, 10, 1, 1, 2, True, False),
])
def test_conv_splitk_vs_unsplit(N, Cin, Cout, H, W, k, stride, ksplit, relu, res):
    """Split-K instances (small maps): same result as the unsplit kernel up to fp32 summation order, 1e-4 vs fp64; bit-repeatable."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5
    b = torch.randn(Cout, device='cuda')
    ref = F.conv2d(x.double(), w.double(), b.double(), stride=stride, padding=k // 2)
    r = torch.randn_like(ref).float() if res else None
    if res:
        ref = ref + r.double()
    if relu:
        ref = ref.clamp_min(0)
    wp, ldw = ops.pack_conv_weight(w)
    out = ops.conv2d_nhwc_splitk(x, wp, ldw, b, Cout, k, stride, k // 2, ksplit, relu=relu, residual=r)
    one = ops.conv2d_nhwc(x, wp, ldw, b, Cout, k, stride, k // 2, relu=relu, residual=r)
    np.testing.assert_allclose(out.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert float((out - one).abs().max()) < 2e-5
    assert torch.equal(out, ops.conv2d_nhwc_splitk(x, wp, ldw, b, Cout, k, stride, k // 2, ksplit, relu=relu, residual=r))


@pytest.mark.parametrize("shapes,Cin,Cout,relu,bias", [
    ([(2, 14, 14)], 256, 256, True, True),                                     # mask-head layer: two 8-row tiles per ROI
    ([(1, 37, 53)], 256, 256, False, True),                                    # ragged in both directions
    ([(1, 64, 96), (1, 32, 48), (1, 16, 24), (1, 8, 12), (1, 4, 6)], 256, 256, True, True),   # RPN: five levels in one launch
    ([(1, 40, 24)], 96, 256, True, False),                                     # three K slabs, no bias
    ([(1, 16, 32)], 512, 512, True, True),                                     # res5 conv2: two 256-channel blocks, 2-row tiles
])
@pytest.mark.parametrize("io", [0, 1, 2, 3])
def test_conv3x3_wreg_bf16_kernel(shapes, Cin, Cout, relu, bias, io):
    """csrc/conv3x3_wreg_bf16.hip (3x3 / 1 / 1, output channels in blocks of 256, plain bf16: weights straight from L2 into the MFMA,
    one barrier per slab) vs float64 on the bf16-rounded operands; every tile height and the general haloed-patch kernel give the
    same bits (same products, same K order); io = bf16 inputs (bit 0) / outputs (bit 1)."""
    from upsnet_amd import ops
    from upsnet_amd._lib import lib
    torch.manual_seed(Cin + len(shapes) + io)
    in16, out16 = bool(io & 1), bool(io & 2)
    xs = [torch.randn(n, Cin, h, w, device='cuda') for n, h, w in shapes]
    xin = [x.bfloat16() if in16 else x for x in xs]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (Cin * 9) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    hi, _, ldw = ops.pack_conv_weight_bf16(w, split=False)
    run = lambda: ops.conv2d_nhwc_bf16_multi(xin, hi, None, ldw, b, Cout, 3, 1, 1, relu=relu, out_dtype=torch.bfloat16 if out16 else torch.float32)
    outs = {}
    try:
        for name, (en, th) in {'halo': (0, 0), 'wreg2n': (1, 1), 'wreg2': (1, 2), 'wreg8': (1, 8), 'wreg16': (1, 16), 'auto': (1, 0)}.items():
            assert lib().upsnet_conv_bf16_tuning(en, th) == 0
            outs[name] = run()
    finally:
        lib().upsnet_conv_bf16_tuning(1, 0)
    for x, o in zip(xs, outs['auto']):
        ref = F.conv2d(x.bfloat16().double(), w.bfloat16().double(), None if b is None else b.double(), padding=1)
        if relu:
            ref = ref.clamp_min(0)
        assert o.shape == ref.shape and o.dtype == (torch.bfloat16 if out16 else torch.float32) and o.permute(0, 2, 3, 1).is_contiguous()
        tol = 2.0 ** -8 if out16 else 1e-4
        np.testing.assert_allclose(o.double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol)
    for name in ('halo', 'wreg2n', 'wreg2', 'wreg8', 'wreg16'):
        for a, o in zip(outs[name], outs['auto']):
            assert torch.equal(a, o), name


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride,relu,res,bias", [
    (1, 64, 256, 40, 56, 1, True, 'bf16', True),        # res2 conv3 + bf16 shortcut
    (1, 64, 256, 40, 56, 1, False, None, True),         # res2 projection
    (2, 256, 128, 41, 57, 2, True, None, True),         # res3 conv1: stride 2, odd map, 128 channels (128 x 128 tiles)
    (1, 256, 512, 41, 57, 2, False, None, False),       # res3 projection, no bias
    (1, 1024, 256, 16, 24, 1, False, 'up32', True),     # FPN lateral: fp32 top-down map through the nearest x2 upsampling
    (1, 512, 256, 32, 48, 1, False, 'up16', True),      # the same with a bf16 top-down map
    (1, 2048, 512, 7, 9, 1, True, 'fp32', True),        # K of 128 k-steps, map smaller than a tile
    (3, 128, 384, 5, 5, 1, True, None, True),           # 384 channels: 128-channel blocks
])
@pytest.mark.parametrize("out16", [True, False])
def test_conv1x1_wreg_bf16_kernel(N, Cin, Cout, H, W, stride, relu, res, bias, out16):
    """csrc/conv1x1_wreg_bf16.hip (1x1 layers with bf16 activations in the bf16 mode: both MFMA operands from global memory, no LDS)
    vs float64 on the bf16-rounded operands, and bit for bit against conv_bf16_kernel (same products, same K order)."""
    from upsnet_amd import ops
    from upsnet_amd._lib import lib
    torch.manual_seed(N + Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda').bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r, up = None, res in ('up32', 'up16')
    if res is not None:
        r = torch.randn(N, Cout, Ho // 2, Wo // 2, device='cuda') if up else torch.randn(N, Cout, Ho, Wo, device='cuda')
        if res in ('bf16', 'up16'):
            r = r.bfloat16()
    ref = F.conv2d(x.double(), w.bfloat16().double(), None if b is None else b.double(), stride=stride)
    if r is not None:
        ref = ref + (F.interpolate(r.double(), scale_factor=2, mode='nearest') if up else r.double())
    if relu:
        ref = ref.clamp_min(0)
    hi, _, ldw = ops.pack_conv_weight_bf16(w, split=False)
    run = lambda: ops.conv2d_nhwc_bf16_multi([x], hi, None, ldw, b, Cout, 1, stride, 0, relu=relu, residuals=None if r is None else [r],
                                             residual_up=up, out_dtype=torch.bfloat16 if out16 else torch.float32)[0]
    try:
        assert lib().upsnet_conv1x1_bf16_tuning(0) == 0
        old = run()
        assert lib().upsnet_conv1x1_bf16_tuning(2) == 0      # (2: every layer the kernel can compute, not only those it is faster on)
        out = run()
    finally:
        lib().upsnet_conv1x1_bf16_tuning(1)
    assert out.shape == ref.shape and out.dtype == (torch.bfloat16 if out16 else torch.float32) and out.permute(0, 2, 3, 1).is_contiguous()
    tol = 2.0 ** -8 if out16 else 1e-4
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol)
    assert torch.equal(out, old)


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,bias,out16", [
    (7, 256, 256, 14, 14, True, True, False),     # the mask head's upsampling layer
    (1, 64, 32, 5, 9, False, False, True),        # one K slab, narrow output, bf16 result
    (3, 128, 96, 6, 6, True, True, False),
])
def test_deconv2x2_bf16_vs_torch(N, Cin, Cout, H, W, relu, bias, out16):
    """upsnet_deconv2x2_nhwc_bf16 (ConvTranspose2d 2x2 / stride 2 as one bf16 GEMM with a scatter epilogue; rcnn.py:132-133 in the
    bf16 mode) vs torch's conv_transpose2d in float64 on the bf16-rounded operands."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, device='cuda').bfloat16()
    w = torch.randn(Cin, Cout, 2, 2, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    ref = F.conv_transpose2d(x.double(), w.bfloat16().double(), None if b is None else b.double(), stride=2)
    if relu:
        ref = ref.clamp_min(0)
    hi, ldw = ops.pack_deconv2x2_weight_bf16(w)
    out = ops.deconv2x2_bf16(x, hi, ldw, b, Cout, relu=relu, out_dtype=torch.bfloat16 if out16 else torch.float32)
    assert out.shape == ref.shape and out.permute(0, 2, 3, 1).is_contiguous()
    tol = 2.0 ** -8 if out16 else 1e-4
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=tol, atol=tol)


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,bias", [
    (100, 256, 256, 14, 14, True, True),       # the mask head's upsampling layer at the benchmark's 100 ROIs (BN = 128 instance)
    (7, 256, 256, 14, 14, True, True),         # few ROIs: ragged last row tile
    (2, 64, 32, 5, 9, False, False),           # BN = 64 instance, rows wrap inside a 32-row block, no bias
    (1, 96, 64, 3, 3, True, True),             # fewer rows than one tile
])
def test_deconv2x2_frag_vs_torch(N, Cin, Cout, H, W, relu, bias):
    """upsnet_deconv2x2_frag_nhwc_f32 (ConvTranspose2d 2x2 / stride 2 on the lean fp32 GEMM kernel with a scatter epilogue,
    csrc/conv1x1.hip MODE 2; rcnn.py:132-133) vs torch's conv_transpose2d in float64 at rtol = atol = 1e-4, and == the general kernel's
    result up to the summation order (both exact-product fp32 MFMA)."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin + Cout)
    x = torch.randn(N, Cin, H, W, device='cuda')
    w = torch.randn(Cin, Cout, 2, 2, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    ref = F.conv_transpose2d(x.double(), w.double(), None if b is None else b.double(), stride=2)
    if relu:
        ref = ref.clamp_min(0)
    wp, b4 = ops.pack_deconv2x2_weight_frag(w, b)
    out = ops.deconv2x2_frag(x, wp, b4, Cout, relu=relu)
    assert out.shape == ref.shape and out.permute(0, 2, 3, 1).is_contiguous()
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    wq, ldw = ops.pack_deconv2x2_weight(w)
    old = ops.deconv2x2(x, wq, ldw, b, Cout, relu=relu)
    np.testing.assert_allclose(out.cpu().numpy(), old.cpu().numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu", [(17, 256, 256, 14, 14, True), (3, 64, 96, 9, 11, False), (1, 128, 64, 32, 40, True)])
def test_winograd_32_channel_form_is_bit_identical(N, Cin, Cout, H, W, relu):
    """upsnet_conv2d_winograd_nhwc_f32_tn32 (32-tile x 32-channel workgroups for any Cout, own weight order) == the 32 x 64 form, bit for
    bit, and 1e-4 vs float64; writes into a caller-provided batch slice."""
    from upsnet_amd import ops
    torch.manual_seed(N + Cin)
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device='cuda')
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = ref.clamp_min(0) if relu else ref
    wp, ldw = ops.pack_winograd_weight(w)
    wp32, ldw32 = ops.pack_winograd_weight(w, tn32=True)
    a = ops.conv2d_winograd_multi([x], wp, ldw, b, Cout, relu=relu)[0]
    whole = ops._nhwc_out(N + 2, Cout, H, W, x.device)
    whole.fill_(-7.0)
    got = ops.conv2d_winograd_multi([x], wp32, ldw32, b, Cout, relu=relu, outs=[whole[1:N + 1]], tn32=True)[0]
    assert torch.equal(got, a) and got.data_ptr() == whole[1:].data_ptr()
    assert bool((whole[0] == -7.0).all()) and bool((whole[N + 1] == -7.0).all())       # nothing outside the slice is touched
    np.testing.assert_allclose(a.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("fused", [True, False])
def test_winograd_tail_split_of_a_batched_launch(fused, monkeypatch):
    """hipconv: 100 ROIs of 14 x 14 on the 32 x 64 form are 616 workgroups = 1.2 rounds; the first 80 ROIs stay on it,
    the last 20 (one half-size workgroup per CU) run on the 32-channel form -- in the same launch (conv_wino16_tail_f32_kernel) or as a second one -- the same bits as the
    unsplit launch, for every ROI."""
    from upsnet_amd import ops
    from upsnet_amd.models import hipconv
    torch.manual_seed(3)
    m = torch.nn.Conv2d(256, 256, 3, 1, 1).cuda()
    x = torch.randn(100, 256, 14, 14, device='cuda').contiguous(memory_format=torch.channels_last)
    monkeypatch.setattr(hipconv, 'WINO_TAIL_SPLIT', True)
    monkeypatch.setattr(hipconv, 'WINO_TAIL_FUSED', fused)
    monkeypatch.setattr(hipconv, 'WINO36_ROI', False)
    assert hipconv._wino_tail_split(m, x) == 80 and hipconv._wino_tail_split(m, x[:64]) == 0
    hipconv.TRACE = []
    try:
        with torch.no_grad():
            y = hipconv.conv(m, x, relu=True, winograd='always')
            form, kform = hipconv.TRACE[-1]['form'], ops.last_kernel_form()
            hipconv.WINO_TAIL_SPLIT = False
            y0 = hipconv.conv(m, x, relu=True, winograd='always')
    finally:
        hipconv.TRACE = None
    assert form == 'winograd tm32 + tail tn32' and torch.equal(y, y0)
    assert kform == ('wino_tail<512,256>' if fused else 'wino<0,32,32>')


@pytest.mark.parametrize("N,n_main,C,Cout,H,W,relu", [(7, 3, 32, 64, 14, 14, True), (5, 4, 64, 192, 9, 11, False), (100, 83, 256, 256, 14, 14, True)])
def test_winograd_tail_entry_point(N, n_main, C, Cout, H, W, relu):
    """upsnet_conv2d_winograd_nhwc_f32_tail: images [0, n_main) on 32 x 64 workgroups and the rest on 32 x 32 workgroups in one launch ==
    the plain launch bit for bit, 1e-4 vs float64; odd maps, ragged tiles, Cout not a multiple of 128."""
    from upsnet_amd import ops
    torch.manual_seed(N + C)
    x = torch.randn(N, C, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, C, 3, 3, device='cuda') / (3 * C ** 0.5)
    b = torch.randn(Cout, device='cuda')
    wp, ldw = ops.pack_winograd_weight(w)
    wp32, ldw32 = ops.pack_winograd_weight(w, tn32=True)
    y = ops.conv2d_winograd_tail(x, wp, ldw, wp32, ldw32, b, Cout, n_main, relu=relu)
    y0 = ops.conv2d_winograd_multi([x], wp, ldw, b, Cout, relu=relu)[0]
    assert torch.equal(y, y0)
    ref = F.conv2d(x.double(), w.double(), b.double(), padding=1)
    ref = ref.clamp_min(0) if relu else ref
    np.testing.assert_allclose(y.cpu().numpy(), ref.float().cpu().numpy(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("N,Cin,Cout,H,W,ks,relu,res", [
    (1, 1024, 256, 50, 84, 4, True, False),     # UPSNet-101-DCN res4 conv1 at 800x1333: 264 tiles
    (1, 1024, 256, 5, 21, 11, True, False),     # the tail rows of that layer alone (104 rows), split 11 ways
    (1, 2048, 512, 25, 42, 11, True, False),    # res5 conv1: less than one round
    (2, 256, 1024, 13, 9, 2, True, True),       # BN = 128 instance... with a residual, ragged last tile, two images
    (1, 96, 64, 7, 5, 3, False, True),          # odd number of K steps (3), one tile
])
def test_conv1x1_frag_splitk(N, Cin, Cout, H, W, ks, relu, res):
    """upsnet_conv1x1_frag_nhwc_f32_splitk (lean 1x1 kernel MODE 3 + reduce): 1e-4 vs float64, within fp32 summation order of the unsplit
    kernel, bit-repeatable; writes through a caller-provided output view."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + H)
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda')
    r = torch.randn(N, Cout, H, W, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    ref = F.conv2d(x.double(), w.double(), b.double()) + (r.double() if res else 0)
    ref = ref.clamp_min(0) if relu else ref
    wp = ops.pack_conv1x1_weight(w)
    one = ops.conv1x1_frag(x, wp, b, Cout, 1, relu=relu, residual=r)
    out = ops.conv1x1_frag(x, wp, b, Cout, 1, relu=relu, residual=r, ksplit=ks)
    np.testing.assert_allclose(out.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert float((out - one).abs().max()) < 2e-5
    assert torch.equal(out, ops.conv1x1_frag(x, wp, b, Cout, 1, relu=relu, residual=r, ksplit=ks))


@pytest.mark.parametrize("N,Cin,Cout,H,W,stride,relu,res,bias", [
    (1, 1024, 256, 50, 84, 1, True, False, True),     # UPSNet-101-DCN res4 conv1 at 800x1333 (4200 pixels: ragged last tile for every tile size)
    (1, 256, 1024, 50, 84, 1, True, True, True),      # res4 conv3 + residual
    (2, 128, 512, 13, 9, 1, False, True, False),      # two images, 234 pixels, no bias
    (1, 512, 256, 27, 43, 2, True, False, True),      # stride 2, odd map
    (1, 32, 64, 7, 5, 1, False, False, True),         # Cin = 32: two 16-channel steps for four waves (two waves walk nothing)
    (1, 80, 36, 6, 6, 1, True, True, True),           # Cin % 32 != 0 (5 steps), Cout % 16 != 0 (padded column blocks), Cout % 4 == 0
    (3, 2048, 128, 5, 5, 1, True, False, True),       # long K walk, 75 pixels
])
def test_conv1x1_ksw_vs_fp64(N, Cin, Cout, H, W, stride, relu, res, bias):
    """csrc/conv1x1_ksw.hip (r13: 16x16x4 MFMA fragments, K split over the four waves of a workgroup, no LDS in the K loop): every tile
    (K split over the waves: 16x64, 32x32, 32x64, 64x64; N split: 16x256, 32x128, 32x256) within rtol = atol = 1e-4 of float64, bit-repeatable, bit-identical for a pixel whatever the batch it
    is launched in (the tile decides the summation order, not the batch), and within fp32 summation-order distance of the 64-pixel kernel."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 1, 1, device='cuda') / Cin ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(N, Cout, Ho, Wo, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), stride=stride) + (r.double() if res else 0)
    ref = ref.clamp_min(0) if relu else ref
    wk = ops.pack_conv1x1_ksw_weight(w)
    frag = ops.conv1x1_frag(x, ops.pack_conv1x1_weight(w), b, Cout, stride, relu=relu, residual=r) if (Cin % 32 == 0 and Cout >= 32) else None
    for tile, sn in [((16, 64), 0), ((32, 32), 0), ((32, 64), 0), ((64, 64), 0), ((16, 256), 1), ((32, 128), 1), ((32, 256), 1)]:
        y = ops.conv1x1_ksw(x, wk, b, Cout, tile, stride=stride, relu=relu, residual=r, split_n=sn)
        assert ops.last_kernel_form() == 'conv1x1_ksw<%d,%d,%s>' % (tile + ('n' if sn else 'k',))
        np.testing.assert_allclose(y.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4, err_msg=str(tile))
        assert torch.equal(y, ops.conv1x1_ksw(x, wk, b, Cout, tile, stride=stride, relu=relu, residual=r, split_n=sn))
        y1 = ops.conv1x1_ksw(x[N - 1:], wk, b, Cout, tile, stride=stride, relu=relu, residual=None if r is None else r[N - 1:], split_n=sn)
        assert torch.equal(y[N - 1:], y1), tile
        if frag is not None:
            assert float((y - frag).abs().max()) < 5e-5
    with pytest.raises(RuntimeError):
        ops.conv1x1_ksw(x, wk, b, Cout, (48, 64), stride=stride, relu=relu, residual=r)


@pytest.mark.parametrize("N,Cin,Cout,H,W,relu,bias", [
    (1, 256, 18, 50, 84, False, True),      # UPSNet-101-DCN res4 offset predictor at 800x1333 (263 tiles: 4 waves per tile)
    (1, 512, 18, 25, 42, False, True),      # res5 (66 tiles: 16 waves)
    (1, 128, 18, 21, 77, True, True),       # 102 tiles: 8 waves; ReLU
    (2, 64, 32, 9, 7, False, False),        # two images (taps must not cross into the neighbouring image), all 32 columns, no bias
    (1, 48, 5, 3, 3, True, True),           # 3 channel chunks per tap, one ragged tile
    (3, 16, 27, 1, 5, False, True),         # H = 1: only the middle row of taps is inside
])
def test_conv3x3_ksw_vs_fp64(N, Cin, Cout, H, W, relu, bias):
    """csrc/conv1x1_ksw.hip, conv3x3_ksw_f32_kernel (r13): 3x3 / stride 1 / pad 1 into <= 32 channels on 16-pixel tiles, the (tap, channel)
    walk split over 4 / 8 / 16 waves: rtol = atol = 1e-4 vs float64, bit-repeatable; the last image alone gives the same values up to the
    summation tree of its wave count."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + Cout + H)
    x = torch.randn(N, Cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') / (9 * Cin) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    ref = ref.clamp_min(0) if relu else ref
    wk = ops.pack_conv3x3_ksw_weight(w)
    y = ops.conv3x3_ksw(x, wk, b, Cout, relu=relu)
    assert ops.last_kernel_form().startswith('conv3x3_ksw<16,32,')
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    assert torch.equal(y, ops.conv3x3_ksw(x, wk, b, Cout, relu=relu))
    y1 = ops.conv3x3_ksw(x[N - 1:], wk, b, Cout, relu=relu)     # (a different wave count may be picked for the smaller launch: then only 1e-4-close)
    np.testing.assert_allclose(y[N - 1:].cpu().numpy(), y1.cpu().numpy(), rtol=1e-5, atol=1e-5)


def test_conv1x1_balanced_main_plus_tail():
    """hipconv: the 1024 -> 256 layer on the 50 x 84 map (66 x 4 = 264 tiles on 256 CUs) runs as 4096 unsplit rows + 104 split-K rows into
    ONE output tensor; 1e-4 vs float64, and equal to the plain launch on the unsplit rows."""
    from upsnet_amd.models import hipconv
    torch.manual_seed(5)
    m = torch.nn.Conv2d(1024, 256, 1).cuda()
    x = torch.randn(1, 1024, 50, 84, device='cuda').contiguous(memory_format=torch.channels_last)
    was, hipconv.BALANCE = hipconv.BALANCE, True       # (opt-in: UPSNET_CONV1X1_BALANCE=1)
    was_ksw, hipconv.KSW = hipconv.KSW, False          # (r13: this layer goes to the small-tile kernel by default -- the balanced form is what the 64-pixel kernel does without it)
    rows, ks = hipconv._c1_balance(m, x)
    assert rows == 4096 and ks > 1
    hipconv.TRACE = []
    try:
        with torch.no_grad():
            y = hipconv.conv(m, x, relu=True)
            form = hipconv.TRACE[-1]['form']
            hipconv.BALANCE = False
            y0 = hipconv.conv(m, x, relu=True)
    finally:
        hipconv.TRACE, hipconv.BALANCE, hipconv.KSW = None, was, was_ksw
    assert form.startswith('conv1x1 main + tail splitk') and y.shape == y0.shape
    ref = F.relu(F.conv2d(x.double(), m.weight.double(), m.bias.double()))
    np.testing.assert_allclose(y.double().cpu().numpy(), ref.detach().cpu().numpy(), rtol=1e-4, atol=1e-4)
    flat, flat0 = y.permute(0, 2, 3, 1).reshape(-1, 256), y0.permute(0, 2, 3, 1).reshape(-1, 256)
    assert torch.equal(flat[:4096], flat0[:4096]) and float((flat[4096:] - flat0[4096:]).abs().max()) < 2e-5


@pytest.mark.parametrize("segs,Cin,Cout,relu,bias", [
    ([(1, 40, 56)], 64, 64, True, True),            # tiles divide the map
    ([(1, 33, 47)], 256, 256, False, True),         # ragged: partial 4x4 tiles on both edges, last m-tile partly empty
    ([(2, 16, 24)], 128, 128, True, False),         # batch, no bias
    ([(1, 20, 20)], 256, 100, True, True),          # Cout not a multiple of 64 (padded columns)
    ([(7, 14, 14)], 256, 256, True, True),          # the mask head's ROI maps (3.5 tiles per side)
    ([(1, 64, 128), (1, 32, 64), (1, 16, 32), (1, 8, 16), (1, 4, 8)], 256, 256, True, True),   # five maps in one launch (the RPN convolution)
    ([(1, 3, 5)], 32, 64, False, True),             # a map smaller than one tile
    ([(5, 1, 9)], 32, 64, False, True),             # H = 1, several images (r13: `risky` from the real extent -- the old rule let image N - 3 read
    ([(3, 2, 7)], 32, 64, True, True),              #   past the tensor); H = 2; H = 3: 35 / 6 / 8 tiles, the last rows of the tensor within 5 rows
    ([(4, 3, 6)], 64, 64, True, False),             #   of most tiles
    ([(2, 70, 93)], 32, 64, True, True),            # 2 x 432 tiles: 25 of the 27 workgroups on the scalar-offset loads (border tiles zeroed after
                                                    # the load, image 0's bottom patches read into image 1), the last two on the bounds-flagged form
])
def test_winograd36_vs_fp64(segs, Cin, Cout, relu, bias):
    """csrc/conv_wino36.hip (Winograd F(4x4,3x3), points {0, +-3/4, +-3/2, inf}, r12) vs torch float64 at rtol = atol = 1e-4 -- the bar of
    the F(2x2) kernel -- on post-ReLU unit-scale activations and He-scaled weights (the model's own layers: tests/test_layerwise_gpu.py),
    with the measured margin asserted (worst error <= 0.35 of the bound), and vs the F(2x2) kernel (same value within 2e-4). Both forms of
    the patch loads are exercised: a workgroup holding tiles of the last two tile rows of the last image flags out-of-image offsets, every
    other one loads unconditionally and zeroes afterwards."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + Cout + len(segs))
    xs = [torch.randn(n, Cin, h, w, device='cuda').relu_().contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    wp, ldw = ops.pack_winograd36_weight(w)
    outs = ops.conv2d_winograd36_multi(xs, wp, ldw, b, Cout, relu)
    assert ops.last_kernel_form() == 'wino36<32,64>'
    w2, ld2 = ops.pack_winograd_weight(w)
    outs2 = ops.conv2d_winograd_multi(xs, w2, ld2, b, Cout, relu)
    worst = 0.0
    for x, o, o2 in zip(xs, outs, outs2):
        ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
        if relu:
            ref = ref.clamp_min(0)
        assert o.shape == ref.shape
        ratio = ((o.double() - ref).abs() / (1e-4 + 1e-4 * ref.abs())).max().item()
        worst = max(worst, ratio)
        np.testing.assert_allclose(o.cpu().numpy(), o2.cpu().numpy(), rtol=2e-4, atol=2e-4)
    assert worst <= 0.35, worst


@pytest.mark.parametrize("name,segs,Cin,Cout", [
    ('fpn_p2', [(1, 256, 512)], 256, 256),                                                   # FPN P2 output convolution at 1024x2048
    ('rpn_x5', [(1, 256 >> l, 512 >> l) for l in range(5)], 256, 256),                        # the five-map RPN launch at 1024x2048
    ('res2_conv2', [(1, 256, 512)], 64, 64),                                                 # res2 conv2
])
def test_winograd36_real_size_margin(name, segs, Cin, Cout):
    """ The F(4x4,3x3) kernel at the REAL sizes hipconv routes to it on the headline workload -- 1x256x256x512 -> 256,
    the five-map RPN launch, res2's 64 -> 64 -- vs torch float64 on post-ReLU unit-scale activations and He-scaled weights, rtol = atol =
    1e-4, with the margin asserted: worst error <= 0.35 of the bound (the figure the small-map test asserts; 131 072-pixel maps take the
    maximum over 100x more outputs than 70x93). The F(2x2) kernel on the same input is asserted <= 0.15 so the 3-4x ratio stays visible."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + len(segs))
    xs = [torch.randn(n, Cin, h, w, device='cuda').relu_().contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda')
    wp, ldw = ops.pack_winograd36_weight(w)
    outs = ops.conv2d_winograd36_multi(xs, wp, ldw, b, Cout, True)
    assert ops.last_kernel_form() == 'wino36<32,64>'
    w2, ld2 = ops.pack_winograd_weight(w)
    outs2 = ops.conv2d_winograd_multi(xs, w2, ld2, b, Cout, True)
    worst, worst2 = 0.0, 0.0
    for x, o, o2 in zip(xs, outs, outs2):
        ref = F.conv2d(x.double(), w.double(), b.double(), padding=1).clamp_min(0)
        bound = 1e-4 + 1e-4 * ref.abs()
        worst = max(worst, ((o.double() - ref).abs() / bound).max().item())
        worst2 = max(worst2, ((o2.double() - ref).abs() / bound).max().item())
        del ref, bound
    print('winograd36 real size %s: F(4x4) worst %.3f of the bound, F(2x2) %.3f' % (name, worst, worst2))
    assert worst <= 0.35 and worst2 <= 0.15, (worst, worst2)


@pytest.mark.parametrize("N,H,W,Cin,Cout,ks,relu,bias", [
    (1, 64, 128, 256, 256, 4, True, True),        # res4 conv2 / FPN P4 at 1024x2048
    (1, 128, 256, 128, 128, 2, True, True),       # res3 conv2
    (1, 33, 47, 64, 100, 3, False, True),         # ragged tiles, Cout % 64 != 0, 4 slabs split 3 ways (1 + 1 + 2)
    (2, 9, 13, 32, 64, 2, True, False),           # two images, one slab each
])
def test_winograd36_splitk_vs_fp64_and_unsplit(N, H, W, Cin, Cout, ks, relu, bias):
    """csrc/conv_wino36.hip, r13: the split-K instance (each workgroup stores the output transform of its partial sums, the reduce kernel adds
    them + bias + ReLU) within 1e-4 of float64 with the margin of the unsplit kernel (<= 0.35), within summation-order distance of the unsplit
    kernel, bit-repeatable."""
    from upsnet_amd import ops
    torch.manual_seed(Cin + Cout + ks)
    x = torch.randn(N, Cin, H, W, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
    w = torch.randn(Cout, Cin, 3, 3, device='cuda') * (2.0 / (9 * Cin)) ** 0.5
    b = torch.randn(Cout, device='cuda') if bias else None
    wp, ldw = ops.pack_winograd36_weight(w)
    y = ops.conv2d_winograd36_splitk(x, wp, ldw, b, Cout, ks, relu=relu)
    assert ops.last_kernel_form() == 'wino36<32,64> splitk%d' % ks
    y0 = ops.conv2d_winograd36_multi([x], wp, ldw, b, Cout, relu)[0]
    ref = F.conv2d(x.double(), w.double(), None if b is None else b.double(), padding=1)
    ref = ref.clamp_min(0) if relu else ref
    worst = ((y.double() - ref).abs() / (1e-4 + 1e-4 * ref.abs())).max().item()
    assert worst <= 0.35, worst
    assert float((y - y0).abs().max()) < 5e-5
    assert torch.equal(y, ops.conv2d_winograd36_splitk(x, wp, ldw, b, Cout, ks, relu=relu))


@pytest.mark.gpu
def test_winograd36_roi_batches_do_not_depend_on_the_batch(monkeypatch):
    """hipconv (r11, UPSNET_WINO36_ROI=1): a pinned layer fed by ROI batches (the mask head) runs on the F(4x4,3x3) kernel for every batch
    size -- a ROI's result has the same bits alone, among 7 and among 100, and is within 1e-4 of float64."""
    from upsnet_amd.models import hipconv
    monkeypatch.setattr(hipconv, 'WINO36_ROI', True)
    torch.manual_seed(5)
    m = torch.nn.Conv2d(256, 256, 3, 1, 1).cuda()
    x = torch.randn(100, 256, 14, 14, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m.weight.mul_(0.3)
        hipconv.TRACE = []
        try:
            y = hipconv.conv(m, x, relu=True, winograd='always')
            forms = {hipconv.TRACE[-1]['form']}
            y7 = hipconv.conv(m, x[40:47].contiguous(memory_format=torch.channels_last), relu=True, winograd='always')
            y1 = hipconv.conv(m, x[99:100].contiguous(memory_format=torch.channels_last), relu=True, winograd='always')
            forms |= {r['form'] for r in hipconv.TRACE[-2:]}
        finally:
            hipconv.TRACE = None
        assert forms == {'winograd36 roi'}, forms
        assert torch.equal(y[40:47], y7) and torch.equal(y[99:100], y1)
        ref = F.relu(F.conv2d(x[:8].double(), m.weight.double(), m.bias.double(), padding=1))
    np.testing.assert_allclose(y[:8].double().cpu().numpy(), ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
