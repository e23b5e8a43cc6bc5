"""GPU parity tests: every HIP op called through the C ABI vs the CPU oracle on the same seeded inputs.
Bit-exact (==) for index / integer / sampled-value outputs; 1e-4 for GEMM-accumulated logits."""
import numpy as np
import pytest
import torch

import oracle
from oracle import ops as oops
from conftest import gen_dets, gen_rois

pytestmark = pytest.mark.gpu


def cu(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


@pytest.fixture(scope="module")
def U():
    from upsnet_amd import ops
    return ops


# ------------------------------------------------------------------ ROIAlign
@pytest.mark.parametrize("C,H,W,ph,scale", [(8, 20, 30, 7, 0.25), (16, 64, 128, 14, 0.125), (4, 5, 7, 7, 1.0 / 32)])
def test_roi_align_nchw_bitexact(U, C, H, W, ph, scale):
    rng = np.random.default_rng(0)
    feat = rng.normal(size=(1, C, H, W)).astype(np.float32)
    rois = gen_rois(rng, 50, int(H / scale), int(W / scale), 4, max(8, int(H / scale)))
    rois = np.vstack([rois, [[0, 0, 0, 0, 0]], [[0, -20, -20, -5, -5]], [[0, W / scale + 50, 3, W / scale + 90, 9]],
                      [[0, 10, 10, 5, 5]]]).astype(np.float32)  # degenerate / outside / inverted
    ref = oracle.roi_align_forward(feat, rois, ph, ph, scale)
    out = U.roi_align_nchw(cu(feat), cu(rois), ph, ph, scale).cpu().numpy()
    assert np.array_equal(out, ref)
    out2 = U.roi_align_nhwc(cu(feat).contiguous(memory_format=torch.channels_last), cu(rois), ph, ph, scale)
    assert np.array_equal(out2.cpu().numpy(), ref)


@pytest.mark.parametrize("variant", [0, 1, 2, 3, 4])
@pytest.mark.parametrize("N,ph,C", [(300, 7, 32), (64, 14, 32), (1, 7, 32), (40, 7, 256), (9, 14, 320), (33, 32, 8)])
def test_fpn_roi_align_bitexact(U, N, ph, C, variant):
    """variant: 0 = LDS tap-table kernel, 1 = its two-register-set form, 2 = the r03-r07 per-bin setup kernel, 3 = the table kernel
    loading each bin's UNIQUE corner cells only (r11; the ROIs here span 4-200 px, so every sharing class of both axes occurs), 4 = 3 with
    packed blend arithmetic -- the same bits from all of them (and from the oracle), incl. C > 256 (two channel passes), C < 256 (idle lanes) and 32x32 bins (table capacity)."""
    from upsnet_amd._lib import lib
    rng = np.random.default_rng(1)
    H, W = 128, 256
    feats = [rng.normal(size=(1, C, H // s, W // s)).astype(np.float32) for s in (4, 8, 16, 32)]
    rois = gen_rois(rng, N, H, W, 4, 200)
    # force boundary cases of the level formula: sqrt(wh)/224 + 1e-6 == 0.5 / 1 / 2
    for i, side in enumerate([112, 224, 448]):
        if i < N:
            rois[i] = [0, 0, 0, side - 1, side - 1]
    # boxes partly / wholly outside the image (empty samples on one or both axes), degenerate boxes
    if N >= 8:
        rois[3] = [0, -40, -30, 20, 25]
        rois[4] = [0, W - 10, H - 12, W + 60, H + 40]
        rois[5] = [0, W + 5, 10, W + 50, 60]
        rois[6] = [0, 17.3, 21.9, 17.3, 21.9]
        rois[7] = [0, -300, -300, -200, -250]
    ref = oops.fpn_roi_align(feats, rois, ph, ph)
    lib().upsnet_roi_tuning(variant)
    try:
        out, lv = U.fpn_roi_align([cu(f) for f in feats], cu(rois), ph, ph, [1 / 4., 1 / 8., 1 / 16., 1 / 32.], return_levels=True)
    finally:
        lib().upsnet_roi_tuning(-1)
    assert np.array_equal(lv.cpu().numpy(), oops.fpn_level(rois))
    assert np.array_equal(out.cpu().numpy(), ref)


def test_fpn_roi_align_padded_tail(U):
    rng = np.random.default_rng(2)
    feats = [rng.normal(size=(1, 8, 64 // s, 64 // s)).astype(np.float32) for s in (4, 8, 16, 32)]
    rois = gen_rois(rng, 10, 64, 64, 4, 60)
    nd = torch.tensor([6], dtype=torch.int32).cuda()
    out = U.fpn_roi_align([cu(f) for f in feats], cu(rois), 7, 7, [1 / 4., 1 / 8., 1 / 16., 1 / 32.], num_rois_dev=nd).cpu().numpy()
    assert np.array_equal(out[:6], oops.fpn_roi_align(feats, rois[:6], 7, 7))
    assert not out[6:].any()


@pytest.mark.parametrize("N,ph,nvalid", [(1000, 7, None), (300, 7, None), (100, 14, None), (1, 7, None), (37, 7, 21), (2048, 7, 1999), (9, 14, 0)])
def test_fpn_roi_align_xcd_order_is_a_permutation_and_changes_no_bit(U, N, ph, nvalid):
    """r13 (csrc/roi_align.hip, fpn_roi_order_kernel): the workgroup -> ROI table that deals the ROIs to the XCDs by image neighbourhood is a
    permutation of 0..N-1 for every N (multiples of 8 or not, a device-side valid count, none valid), rows beyond the valid count sit at
    its end, and the launch through it returns the SAME bits as the launch in ROI order == the oracle -- at the benchmark shapes too."""
    rng = np.random.default_rng(N + ph)
    H, W = 256, 512
    feats = [rng.normal(size=(1, 32, H // s, W // s)).astype(np.float32) for s in (4, 8, 16, 32)]
    rois = gen_rois(rng, N, H, W, 4, 200)
    nd = None if nvalid is None else torch.tensor([nvalid], dtype=torch.int32).cuda()
    g = [cu(f) for f in feats]
    order = U.fpn_roi_order(cu(rois), (H, W), num_rois_dev=nd)
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(N))
    nv = N if nvalid is None else nvalid
    # workgroup b runs on XCD b % 8 and takes ROI o[b]: rows beyond the valid count were ordered last = the last positions of the last XCDs
    cnt = [(N - j + 7) // 8 for j in range(8)]
    pos = np.empty(N, np.int64)
    for b in range(N):
        pos[o[b]] = sum(cnt[:b % 8]) + b // 8
    if nv < N:
        assert pos[nv:].min() >= nv and (nv == 0 or pos[:nv].max() < nv)
    sc = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    a = U.fpn_roi_align(g, cu(rois), ph, ph, sc, num_rois_dev=nd, order=None)
    b = U.fpn_roi_align(g, cu(rois), ph, ph, sc, num_rois_dev=nd, order=order)
    c = U.fpn_roi_align(g, cu(rois), ph, ph, sc, num_rois_dev=nd)            # 'auto'
    assert torch.equal(a, b) and torch.equal(a, c)
    if N <= 300:
        assert not a.cpu().numpy()[nv:].any()
        if nv:
            assert np.array_equal(a.cpu().numpy()[:nv], oops.fpn_roi_align(feats, rois[:nv], ph, ph))
    if nv >= 64:   # the dealing does what it is for: the ROIs of one XCD are neighbours (same level, nearby stripes) -- fewer distinct
        lv = oops.fpn_level(rois[:nv])            # (level, stripe) cells per XCD than in ROI order
        stripe = np.clip(((rois[:nv, 2] + rois[:nv, 4]) * 0.5 * 16 / H).astype(np.int64), 0, 15)
        cell = lv * 16 + stripe
        dealt = np.mean([len(set(cell[[o[b] for b in range(j, N, 8) if o[b] < nv]])) for j in range(8)])
        plain = np.mean([len(set(cell[[b for b in range(j, N, 8) if b < nv]])) for j in range(8)])
        assert dealt < 0.6 * plain, (dealt, plain)


# ------------------------------------------------------------------ deformable conv
@pytest.mark.parametrize("C,H,W,k,pad,stride,dil,dg", [(8, 12, 17, 3, 1, 1, 1, 1), (12, 9, 9, 3, 2, 1, 2, 2), (4, 16, 16, 3, 1, 2, 1, 1)])
def test_deform_im2col_bitexact(U, C, H, W, k, pad, stride, dil, dg):
    rng = np.random.default_rng(3)
    im = rng.normal(size=(C, H, W)).astype(np.float32)
    Ho, Wo = U.out_hw(H, W, (k, k), (pad, pad), (stride, stride), (dil, dil))
    off = (rng.normal(size=(dg * 2 * k * k, Ho, Wo)) * 2).astype(np.float32)
    off[:, 0, 0] = 0.0
    off[0, 1, 1] = -1.0 - pad  # lands exactly on the -1 border
    mask = rng.uniform(0, 2, size=(dg * k * k, Ho, Wo)).astype(np.float32)
    ref = oracle.deform_im2col(im, off, (k, k), (pad, pad), (stride, stride), (dil, dil), dg)
    col = torch.zeros((C * k * k, Ho, Wo), device='cuda')
    U.deform_im2col(cu(im), cu(off), (1, C, H, W), tuple(col.shape), (k, k), (pad, pad), (stride, stride), (dil, dil), 1, dg, col)
    assert np.array_equal(col.cpu().numpy(), ref)
    ref2 = oracle.deform_im2col(im, off, (k, k), (pad, pad), (stride, stride), (dil, dil), dg, mask=mask)
    col.zero_()
    U.mod_deform_im2col(cu(im), cu(off), cu(mask), (1, C, H, W), tuple(col.shape), (k, k), (pad, pad), (stride, stride), (dil, dil), dg, col)
    assert np.array_equal(col.cpu().numpy(), ref2)


def _dcn_ref(x, off, w, b, k, pad, stride, dil, mask=None, relu=False):
    col = oracle.deform_im2col(x, off, (k, k), (pad, pad), (stride, stride), (dil, dil), 1, mask=mask)
    out = (w.reshape(w.shape[0], -1).astype(np.float64) @ col.reshape(col.shape[0], -1).astype(np.float64))
    out = out.reshape(w.shape[0], col.shape[1], col.shape[2])
    if b is not None:
        out = out + b[:, None, None]
    return np.maximum(out, 0) if relu else out


@pytest.mark.parametrize("kind", ["frag", "igemm"])
@pytest.mark.parametrize("cin,cout,sizes,mod,relu", [(32, 32, [(9, 13)], False, False), (64, 128, [(16, 24), (8, 12), (4, 6), (2, 3)], False, True),
                                                     (32, 64, [(10, 10)], True, False), (32, 256, [(7, 19), (3, 5)], False, False),
                                                     (256, 128, [(40, 64), (20, 32)], False, True), (64, 19, [(11, 13)], True, False),
                                                     (128, 160, [(33, 17)], False, False)])
def test_deform_conv_fused_vs_oracle(U, cin, cout, sizes, mod, relu, kind):
    """Both generations of the fused kernel: 'frag' = csrc/deform_fused.hip (default), 'igemm' = the loader mode of csrc/conv.hip."""
    rng = np.random.default_rng(4)
    w = (rng.normal(size=(cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    xs = [rng.normal(size=(1, cin, h, ww)).astype(np.float32) for h, ww in sizes]
    offs = [(rng.normal(size=(1, 18, h, ww)) * 2).astype(np.float32) for h, ww in sizes]
    offs[0][0, :, 0, 0] = [-1.0, -1.0, -1.5, 0.0, 0.0, -2.0, 0.25, float(sizes[0][1]), 1.0, 1.0, 0.5, -0.5, 0.0, 0.0, 2.0, 2.0, float(sizes[0][0]), 0.0]
    masks = [rng.uniform(0, 2, size=(1, 9, h, ww)).astype(np.float32) for h, ww in sizes] if mod else None
    wp = U.pack_dcn_weight(cu(w), kind)
    if kind == 'igemm':
        assert wp[1] == (cout + 31) // 32 * 32
        assert np.array_equal(wp[0].cpu().numpy()[:, :cout], w.transpose(2, 3, 1, 0).reshape(9 * cin, cout))
    else:
        assert wp[0] == 'frag' and wp[1].numel() == (cout + 127) // 128 * 128 * cin * 9
    outs = U.deform_conv_fused([cu(x) for x in xs], [cu(o) for o in offs], wp, cu(b), cin, cout, (3, 3), (1, 1), (1, 1), (1, 1),
                               masks=[cu(m) for m in masks] if mod else None, relu=relu)
    for i, o in enumerate(outs):
        ref = _dcn_ref(xs[i][0], offs[i][0], w, b, 3, 1, 1, 1, masks[i][0] if mod else None, relu)
        got = o.cpu().numpy()[0]
        assert got.shape == ref.shape
        np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)  # fp32 logits within 1e-4 (BASELINE.json)


@pytest.mark.parametrize("cin,cout,sizes,mod,relu,k,pad,dil", [
    (64, 128, [(16, 24), (8, 12), (4, 6), (2, 3)], False, True, 3, 1, 1), (32, 64, [(10, 10)], True, False, 3, 1, 1),
    (256, 128, [(40, 64), (20, 32)], False, True, 3, 1, 1), (64, 19, [(11, 13)], True, False, 3, 2, 2),
    (96, 160, [(33, 17)], False, False, 1, 0, 1), (32, 32, [(9, 9)], False, False, 5, 2, 1)])
def test_deform_conv_fused_bf16_matrix_cores(U, cin, cout, sizes, mod, relu, k, pad, dil):
    """csrc/deform_fused_bf16.hip (BASELINE configs[2]: blended samples and weights rounded to bf16, exact products, fp32 sums) vs
    the oracle evaluated on bf16-rounded weights: what remains is the rounding of the samples -- a few 1e-3 of the output scale."""
    rng = np.random.default_rng(7)
    w = (rng.normal(size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    xs = [rng.normal(size=(1, cin, h, ww)).astype(np.float32) for h, ww in sizes]
    offs = [(rng.normal(size=(1, 2 * k * k, h, ww)) * 2).astype(np.float32) for h, ww in sizes]
    masks = [rng.uniform(0, 2, size=(1, k * k, h, ww)).astype(np.float32) for h, ww in sizes] if mod else None
    wp = U.pack_dcn_weight(cu(w), 'frag_bf16')
    assert wp[0] == 'frag_bf16' and wp[1].dtype == torch.bfloat16 and wp[1].numel() == (cout + 127) // 128 * 128 * cin * k * k
    outs = U.deform_conv_fused([cu(x) for x in xs], [cu(o) for o in offs], wp, cu(b), cin, cout, (k, k), (1, 1), (pad, pad), (dil, dil),
                               masks=[cu(m) for m in masks] if mod else None, relu=relu)
    wr = torch.from_numpy(w).to(torch.bfloat16).float().numpy()
    for i, o in enumerate(outs):
        ref = _dcn_ref(xs[i][0], offs[i][0], wr, b, k, pad, 1, dil, masks[i][0] if mod else None, relu)
        got = o.cpu().numpy()[0]
        assert got.shape == ref.shape
        scale = float(np.abs(ref).max())
        assert float(np.abs(got - ref).max()) <= 1.5e-2 * scale
        assert float(np.sqrt(np.mean((got - ref) ** 2))) <= 3e-3 * scale


@pytest.mark.parametrize("cin,cout,H,W", [(256, 256, 25, 42), (128, 128, 50, 84), (512, 512, 13, 21)])
def test_deform_conv_fused_splitk_small_maps(U, cin, cout, H, W):
    """The DCN bottlenecks of the R101-DCN backbone (configs[3]) are single small maps: the fused kernel splits K over up to 8
    workgroups per tile + a fixed-order reduce. Same result as the unsplit kernel within fp32 summation order, 1e-4 vs the oracle."""
    rng = np.random.default_rng(cin + H)
    w = (rng.normal(size=(cout, cin, 3, 3)) / np.sqrt(cin * 9)).astype(np.float32)
    b = rng.normal(size=(cout,)).astype(np.float32)
    x = rng.normal(size=(1, cin, H, W)).astype(np.float32)
    off = (rng.normal(size=(1, 18, H, W)) * 2).astype(np.float32)
    wp = U.pack_dcn_weight(cu(w), 'frag')
    ks = U.dcn_ksplit([torch.empty(1, cout, H, W)], cin, cout, 9)
    assert ks > 1
    out = U.deform_conv_fused([cu(x)], [cu(off)], wp, cu(b), cin, cout, (3, 3), (1, 1), (1, 1), (1, 1), relu=True)[0].cpu().numpy()[0]
    ref = _dcn_ref(x[0], off[0], w, b, 3, 1, 1, 1, None, True)
    np.testing.assert_allclose(out, ref, rtol=1e-4, atol=1e-4)
    U.DCN_SPLITK = False
    try:
        out1 = U.deform_conv_fused([cu(x)], [cu(off)], wp, cu(b), cin, cout, (3, 3), (1, 1), (1, 1), (1, 1), relu=True)[0].cpu().numpy()[0]
    finally:
        U.DCN_SPLITK = True
    np.testing.assert_allclose(out, out1, rtol=1e-5, atol=1e-5)
    again = U.deform_conv_fused([cu(x)], [cu(off)], wp, cu(b), cin, cout, (3, 3), (1, 1), (1, 1), (1, 1), relu=True)[0].cpu().numpy()[0]
    assert np.array_equal(out, again)     # fixed-order reduce: bit-repeatable


@pytest.mark.parametrize("k,pad,stride,dil", [(3, 2, 1, 2), (3, 1, 2, 1), (1, 0, 1, 1), (5, 2, 1, 1)])
def test_deform_conv_fused_geometries(U, k, pad, stride, dil):
    """Dilated (the reference's dilated res5 option), strided, 1x1 and 5x5 deformable kernels through csrc/deform_fused.hip; every
    variant of its register-set / occupancy knob computes the same bits."""
    rng = np.random.default_rng(k * 10 + dil)
    cin, cout, H, W = 64, 96, 21, 30
    w = (rng.normal(size=(cout, cin, k, k)) / np.sqrt(cin * k * k)).astype(np.float32)
    x = rng.normal(size=(1, cin, H, W)).astype(np.float32)
    Ho, Wo = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1, (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    off = (rng.normal(size=(1, 2 * k * k, Ho, Wo)) * 1.5).astype(np.float32)
    wp = U.pack_dcn_weight(cu(w), 'frag')
    ref = _dcn_ref(x[0], off[0], w, None, k, pad, stride, dil)
    from upsnet_amd._lib import lib
    outs = []
    try:
        for variant in (1, 2, 5, 6):
            lib().upsnet_dcn_tuning(variant)
            outs.append(U.deform_conv_fused([cu(x)], [cu(off)], wp, None, cin, cout, (k, k), (stride, stride), (pad, pad), (dil, dil))[0].cpu().numpy()[0])
    finally:
        lib().upsnet_dcn_tuning(0)
    np.testing.assert_allclose(outs[0], ref, rtol=1e-4, atol=1e-4)
    for o in outs[1:]:
        assert np.array_equal(o, outs[0])


def test_deform_conv_function_and_module(U):
    from upsnet_amd.operators.modules.deform_conv import DeformConvWithOffset
    torch.manual_seed(0)
    for cin, cout in [(32, 64), (8, 8)]:  # fused path, im2col+mm path
        m = DeformConvWithOffset(cin, cout, kernel_size=3, padding=1).cuda()
        with torch.no_grad():
            m.conv_offset.weight.normal_(0, 0.05)
        x = torch.randn(2, cin, 11, 14, device='cuda')
        with torch.no_grad():
            y = m(x)
            off = m.conv_offset(x)
        for i in range(2):
            ref = _dcn_ref(x[i].cpu().numpy(), off[i].cpu().numpy(), m.conv.weight.detach().cpu().numpy(),
                           m.conv.bias.detach().cpu().numpy(), 3, 1, 1, 1)
            np.testing.assert_allclose(y[i].cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    with pytest.raises(Exception):
        m.cpu()(torch.randn(1, 8, 4, 4))


def test_deform_conv_module_with_non_square_padding(U):
    """DeformConv(..., padding=(a, b)) with a != b must keep working -- the fused kernels take square geometry only, so
    such a layer runs the reference's own structure (HIP im2col into a column buffer + one GEMM, functions/deform_conv.py:44-56)."""
    from upsnet_amd.operators.modules.deform_conv import DeformConv
    torch.manual_seed(1)
    m = DeformConv(32, 48, 3, stride=1, padding=(1, 2), dilation=1, bias=True).cuda()
    x = torch.randn(1, 32, 12, 15, device='cuda')
    Ho, Wo = 12, 17
    off = torch.randn(1, 18, Ho, Wo, device='cuda') * 1.5
    with torch.no_grad():
        y = m(x, off)
    assert y.shape == (1, 48, Ho, Wo)
    col = oracle.deform_im2col(x[0].cpu().numpy(), off[0].cpu().numpy(), (3, 3), (1, 2), (1, 1), (1, 1), 1)
    w = m.weight.detach().cpu().numpy().astype(np.float64).reshape(48, -1)
    ref = (w @ col.reshape(col.shape[0], -1).astype(np.float64)).reshape(48, Ho, Wo) + m.bias.detach().cpu().numpy()[:, None, None]
    np.testing.assert_allclose(y[0].cpu().numpy(), ref, rtol=1e-4, atol=1e-4)
    assert not U.fused_dcn_supported(32, 48, 1, 1, (1, 2), (1, 1), (1, 1)) and U.fused_dcn_supported(32, 48, 1, 1, (1, 1), (1, 1), (1, 1))


def test_zero_fill_kernel_handles_any_address_and_size():
    """ups_zero_async never falls back to a memset node -- unaligned head / tail bytes are written by the same kernel."""
    import ctypes
    from upsnet_amd._lib import lib, stream
    if not hasattr(lib(), 'upsnet_zero_fill'):
        pytest.skip('no test entry point')
    buf = torch.full((4096,), 0xAB, dtype=torch.uint8, device='cuda')
    for start, n in [(0, 0), (1, 1), (3, 2), (1, 7), (2, 1021), (5, 3), (0, 4096), (7, 4000)]:
        buf.fill_(0xAB)
        assert lib().upsnet_zero_fill(stream(), ctypes.c_void_p(buf.data_ptr() + start), ctypes.c_size_t(n)) == 0
        host = buf.cpu().numpy()
        assert (host[start:start + n] == 0).all() and (host[:start] == 0xAB).all() and (host[start + n:] == 0xAB).all(), (start, n)


# ------------------------------------------------------------------ NMS
@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 300, 1000, 1500])
@pytest.mark.parametrize("thresh", [0.5, 0.7])
def test_gpu_nms_bitexact(U, n, thresh):
    rng = np.random.default_rng(n)
    d = gen_dets(rng, n)
    ref = oops.gpu_nms(d, thresh)
    got = U.gpu_nms(cu(d), thresh).cpu().numpy()
    assert np.array_equal(got, ref)
    # `_nms` drop-in on host pointers, sorted input
    order = oops.argsort_desc(d[:, 4])
    keep = U.nms_host(d[order], thresh)
    assert np.array_equal(order[keep], ref)


def test_nms_wrappers_and_empty(U):
    from upsnet_amd.nms.nms import gpu_nms_wrapper
    rng = np.random.default_rng(9)
    d = gen_dets(rng, 200)
    assert gpu_nms_wrapper(0.5, 0)(d) == oops.gpu_nms(d, 0.5).tolist()
    assert gpu_nms_wrapper(0.5, 0)(np.zeros((0, 5), np.float32)) == []


def test_nms_batched_with_counts_and_preremoved(U):
    rng = np.random.default_rng(10)
    P, nmax = 5, 1000
    counts = [1000, 777, 64, 1, 0]
    boxes = np.zeros((P, nmax, 4), np.float32)
    scores = np.zeros((P, nmax), np.float32)
    pre = np.zeros((P, nmax), np.uint8)
    refs = []
    for p, c in enumerate(counts):
        d = gen_dets(rng, c) if c else np.zeros((0, 5), np.float32)
        boxes[p, :c], scores[p, :c] = d[:, :4], d[:, 4]
        pre[p, :c] = rng.uniform(size=c) < 0.1
        live = np.where(pre[p, :c] == 0)[0]
        refs.append(live[oops.gpu_nms(d[live], 0.7)] if len(live) else np.zeros((0,), np.int64))
    keep, cnt = U.nms_batched(cu(boxes), cu(scores), cu(np.array(counts, np.int32)), 0.7, cu(pre))
    keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
    for p in range(P):
        assert cnt[p] == len(refs[p])
        assert np.array_equal(keep[p, :cnt[p]], refs[p])


@pytest.mark.parametrize("lds", [0, 1, 2])
def test_nms_scan_both_mask_sources(U, lds):
    """The three forms of the greedy scan -- 0: row-layout scan for <= 1024 boxes (default; the general scan above that), 1: general
    scan on an LDS copy of the mask, 2: general scan reading L2 at every size -- give the same keep lists, for 1 .. 2000 boxes, odd
    and even numbers of 64-blocks, with and without pre-removed boxes, several problems per launch."""
    from upsnet_amd._lib import lib
    rng = np.random.default_rng(5)
    try:
        lib().upsnet_nms_tuning(lds)
        for n in (1, 63, 64, 65, 129, 300, 500, 960, 1000, 1023, 1024, 2000):
            d = gen_dets(rng, n)
            got = U.gpu_nms(cu(d), 0.5).cpu().numpy()
            assert np.array_equal(got, oops.gpu_nms(d, 0.5)), (lds, n)
        # batched: ragged counts, pre-removed flags (the RPN's min-size filter), nmax with an odd number of column blocks
        for P, nmax in ((5, 1000), (8, 300), (3, 1024)):
            counts = rng.integers(0, nmax + 1, P).astype(np.int32)
            counts[0] = nmax
            boxes = np.zeros((P, nmax, 4), np.float32)
            scores = np.zeros((P, nmax), np.float32)
            pre = (rng.uniform(size=(P, nmax)) < 0.1).astype(np.uint8)
            for q in range(P):
                d = gen_dets(rng, nmax)
                boxes[q], scores[q] = d[:, :4], d[:, 4]
            keep, cnt = U.nms_batched(cu(boxes), cu(scores), cu(counts), 0.6, pre_removed=cu(pre))
            keep, cnt = keep.cpu().numpy(), cnt.cpu().numpy()
            for q in range(P):
                c = int(counts[q])
                alive = np.where(pre[q, :c] == 0)[0]
                dq = np.hstack([boxes[q, :c], scores[q, :c, None]])[alive]
                want = alive[oops.gpu_nms(dq, 0.6)] if len(alive) else np.zeros((0,), np.int64)
                assert int(cnt[q]) == len(want) and np.array_equal(keep[q, :len(want)], want), (lds, P, nmax, q)
    finally:
        lib().upsnet_nms_tuning(0)


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("n", [1, 5, 200, 1300])
def test_soft_nms_bitexact(U, method, n):
    rng = np.random.default_rng(100 + n)
    d = gen_dets(rng, n)
    rb, ri = oracle.soft_nms(d, sigma=0.5, Nt=0.3, threshold=0.001 if method else 0.001, method=method)
    gb, gi = U.soft_nms(cu(d), 0.5, 0.3, 0.001, method)
    gi = gi.cpu().numpy()
    assert np.array_equal(gi, ri)
    assert np.array_equal(gb.cpu().numpy()[:len(ri)], rb[:len(ri)])


@pytest.mark.parametrize("method", [0, 1, 2])
@pytest.mark.parametrize("nmax,P", [(1000, 5), (64, 3), (1025, 2), (4096, 2), (4100, 2)])
def test_soft_nms_batched_bitexact(U, method, nmax, P):
    """P problems per launch (RPN levels / classes), ragged counts incl. empty: 1 wavefront (<= 1024 boxes), 4 wavefronts
    (<= 4096, LDS) and the global-memory form (> 4096). The oracle is pinned to the compiled reference (test_ref_cpu_nms.py)."""
    rng = np.random.default_rng(7 * nmax + method)
    counts = [nmax] + [int(c) for c in rng.integers(0, nmax + 1, P - 1)]
    counts[-1] = 0 if P > 2 else counts[-1]
    boxes = np.zeros((P, nmax, 5), np.float32)
    for p, c in enumerate(counts):
        if c:
            boxes[p, :c] = gen_dets(rng, c, ties=(p % 2 == 0))
    thr = 0.001 if method else 0.05
    gb, gi, gn = U.soft_nms_batched(cu(boxes), cu(np.array(counts, np.int32)), 0.5, 0.3, thr, method)
    gb, gi, gn = gb.cpu().numpy(), gi.cpu().numpy(), gn.cpu().numpy()
    for p, c in enumerate(counts):
        rb, ri = oracle.soft_nms(boxes[p, :c], 0.5, 0.3, thr, method)
        assert gn[p] == len(ri), (p, c)
        assert np.array_equal(gi[p, :gn[p]], ri)
        assert np.array_equal(gb[p, :gn[p]].view(np.uint32), rb[:len(ri)].view(np.uint32))


@pytest.mark.parametrize("n", [1, 2, 65, 1000, 3000])
@pytest.mark.parametrize("thresh", [0.3, 0.7])
def test_cpu_nms_ge_bitexact(U, n, thresh):
    """cpu_nms_wrapper semantics (cpu_nms.pyx:77, `>=` against a double threshold) on the device vs the oracle restatement, which
    is pinned to the compiled reference on CPU. Includes pairs whose overlap equals the threshold exactly."""
    from upsnet_amd.nms.nms import cpu_nms_wrapper, py_nms_wrapper
    rng = np.random.default_rng(n)
    d = gen_dets(rng, n, ties=False)
    d[:, 4] = (rng.permutation(n).astype(np.float32) + 1) / (n + 1)
    order = np.argsort(d[:, 4], kind='stable')[::-1]
    assert cpu_nms_wrapper(thresh)(d) == oracle.cpu_nms(d, thresh, order)
    assert py_nms_wrapper(thresh)(d) == [int(order[k]) for k in oracle.nms_sorted(d[order], thresh)]
    a = np.array([[0, 0, 9, 9, 0.9], [0, 5, 9, 14, 0.8]], np.float32)
    t = float(np.float32(50.0) / np.float32(150.0))
    assert cpu_nms_wrapper(t)(a) == [0] and py_nms_wrapper(t)(a) == [0, 1]
    assert cpu_nms_wrapper(float(np.nextafter(t, 1.0)))(a) == [0, 1]


# ------------------------------------------------------------------ proposals
def _rpn_inputs(rng, H, W, strides=(4, 8, 16, 32, 64), A=3, sat=True):
    cls, box = [], []
    for s in strides:
        h, w = max(H // s, 1), max(W // s, 1)
        logit = rng.normal(0, 2.5, size=(1, A, h, w)).astype(np.float32)
        p = (1.0 / (1.0 + np.exp(-logit.astype(np.float64)))).astype(np.float32)
        if sat:
            p[0, :, ::3, ::5] = np.float32(1.0)   # saturated sigmoid => real ties
            p[0, 0, 1::4, 2::7] = p[0, 1, 0, 0]
        cls.append(p)
        box.append((rng.normal(0, 0.5, size=(1, 4 * A, h, w))).astype(np.float32))
    return cls, box


@pytest.mark.parametrize("H,W,pre,post", [(256, 512, 1000, 1000), (64, 96, 1000, 300), (1024, 2048, 1000, 1000), (32, 32, 50, 20)])
def test_pyramid_proposals_bitexact(U, H, W, pre, post):
    from upsnet_amd.operators.modules.pyramid_proposal import PyramidProposal
    rng = np.random.default_rng(H + W)
    cls, box = _rpn_inputs(rng, H, W)
    im_info = np.array([[H - 3, W - 5, 1.0]], np.float32)
    ref_rois, ref_scores = oops.pyramid_proposal(cls, box, im_info[0], pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=0.7)
    pp = PyramidProposal((4, 8, 16, 32, 64), (8,), (0.5, 1, 2), pre, post, 0.7, 0, individual_proposals=True)
    rois, scores = pp([cu(c) for c in cls], [cu(b) for b in box], im_info)
    assert rois.shape[0] == ref_rois.shape[0]
    assert np.array_equal(scores.cpu().numpy(), ref_scores)
    assert np.array_equal(rois.cpu().numpy(), ref_rois)


def test_pyramid_proposals_write_the_roi_xcd_table_of_the_box_head(U):
    """r13: prop_merge_kernel (the launch that ranks the proposals) also writes the workgroup -> ROI table of the box head's ROIAlign launch
    (csrc/roi_order.h): a permutation of 0..post-1 with the rows beyond the kept count at its end, attached to the rois tensor; ops.fpn_roi_align
    picks it up from that tensor ('auto') and returns the bits of the launch in ROI order. The proposals themselves are untouched (== oracle)."""
    from upsnet_amd.operators.functions.pyramid_proposal import PyramidProposalFunction
    H, W, pre, post = 512, 1024, 1000, 1000
    rng = np.random.default_rng(11)
    cls, box = _rpn_inputs(rng, H, W)
    im_info = np.array([[H, W, 1.0]], np.float32)
    ref_rois, ref_scores = oops.pyramid_proposal(cls, box, im_info[0], pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=0.7)
    fn = PyramidProposalFunction((4, 8, 16, 32, 64), (8,), (0.5, 1, 2), pre, post, 0.7, 0, individual_proposals=True)
    rois, scores, num = fn.forward_padded([cu(c) for c in cls], [cu(b) for b in box], cu(im_info[0]))
    k = int(num.item())
    assert np.array_equal(rois[:k].cpu().numpy(), ref_rois) and np.array_equal(scores[:k].cpu().numpy(), ref_scores)
    order = getattr(rois, '_ups_roi_order', None)
    assert order is not None and order.dtype == torch.int32 and order.numel() == post
    o = order.cpu().numpy()
    assert sorted(o.tolist()) == list(range(post))
    cnt = [(post - j + 7) // 8 for j in range(8)]
    pos = np.empty(post, np.int64)
    for b in range(post):
        pos[o[b]] = sum(cnt[:b % 8]) + b // 8
    assert k == post or (pos[k:].min() >= k and pos[:k].max() < k)
    feats = [cu(rng.normal(size=(1, 32, H // s, W // s)).astype(np.float32)) for s in (4, 8, 16, 32)]
    sc = [1 / 4., 1 / 8., 1 / 16., 1 / 32.]
    a = U.fpn_roi_align(feats, rois, 7, 7, sc, num_rois_dev=num, order=None)
    b = U.fpn_roi_align(feats, rois, 7, 7, sc, num_rois_dev=num)                 # 'auto': the attached table
    assert torch.equal(a, b)
    assert np.array_equal(a[:k].cpu().numpy(), oops.fpn_roi_align([f.cpu().numpy() for f in feats], ref_rois, 7, 7))


@pytest.mark.parametrize("H,W,pre,post,thr,min_size", [(256, 512, 6000, 300, 0.7, 0), (64, 96, 1000, 300, 0.7, 4), (1024, 2048, 2000, 1000, 0.7, 16),
                                                       (32, 32, 50, 20, 0.5, 0), (96, 160, 300, 250, 0.3, 8)])
def test_pyramid_proposals_joint_bitexact(U, H, W, pre, post, thr, min_size):
    """individual_proposals=False (the reference constructors' default; functions/pyramid_proposal.py:181-208): joint ranking with real
    score ties (rule (i): higher concatenation index first), size filter BEFORE the top-k, one NMS over up to 6000 boxes, and -- when
    the NMS keeps fewer than post_nms_top_n -- the host-side random padding drawn from numpy's global generator."""
    from upsnet_amd.operators.functions.pyramid_proposal import PyramidProposalFunction
    from upsnet_amd.operators.modules.pyramid_proposal import PyramidProposal
    rng = np.random.default_rng(H + W + pre)
    cls, box = _rpn_inputs(rng, H, W)
    im_info = np.array([[H - 3, W - 5, 1.0]], np.float32)
    # device entry = the kept list before the padding
    ref_rois, ref_scores = oops.pyramid_proposal(cls, box, im_info[0], pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=thr,
                                                 min_size=min_size, individual_proposals=False, pad=False)
    fn = PyramidProposalFunction((4, 8, 16, 32, 64), (8,), (0.5, 1, 2), pre, post, thr, min_size)
    rois, scores, num = fn.forward_padded([cu(c) for c in cls], [cu(b) for b in box], cu(im_info[0]))
    k = int(num.item())
    assert k == len(ref_scores) and k > 0
    assert np.array_equal(scores[:k].cpu().numpy(), ref_scores)
    assert np.array_equal(rois[:k].cpu().numpy(), ref_rois)
    assert not rois[k:].any() and not scores[k:].any()
    # module = padding (same generator state) + stable ranking
    np.random.seed(H + post)
    ref_rois, ref_scores = oops.pyramid_proposal(cls, box, im_info[0], pre_nms_top_n=pre, post_nms_top_n=post, nms_thresh=thr,
                                                 min_size=min_size, individual_proposals=False)
    pp = PyramidProposal((4, 8, 16, 32, 64), (8,), (0.5, 1, 2), pre, post, thr, min_size)
    np.random.seed(H + post)
    rois, scores = pp([cu(c) for c in cls], [cu(b) for b in box], im_info)
    assert tuple(rois.shape) == (post, 1, 5) and tuple(scores.shape) == (post, 1, 1)
    assert np.array_equal(scores.cpu().numpy().reshape(-1), ref_scores)
    assert np.array_equal(rois.cpu().numpy().reshape(-1, 5), ref_rois)


# ------------------------------------------------------------------ detection selection
def _rcnn_inputs(rng, N, C, H, W, peaky):
    rois = gen_rois(rng, N, H, W, 8, 300)
    rois[N // 2:] = rois[:N - N // 2] + np.hstack([np.zeros((N - N // 2, 1)), rng.normal(0, 3, (N - N // 2, 4))]).astype(np.float32)
    delta = rng.normal(0, 0.5, size=(N, 4 * C)).astype(np.float32)
    logit = rng.normal(0, peaky, size=(N, C)).astype(np.float64)
    prob = np.exp(logit - logit.max(1, keepdims=True))
    prob = (prob / prob.sum(1, keepdims=True)).astype(np.float32)
    prob[::11] = prob[3]  # duplicated rows => tied scores
    return rois, delta, prob


@pytest.mark.parametrize("N,C,agn,thresh,peaky", [(1000, 9, False, 0.05, 1.0), (1000, 9, True, 0.6, 3.0), (300, 81, False, 0.05, 2.0),
                                                  (300, 81, True, 0.6, 4.0), (50, 9, True, 0.999, 0.1), (7, 9, False, 0.05, 1.0)])
def test_mask_roi_bitexact(U, N, C, agn, thresh, peaky):
    from upsnet_amd.config.config import config
    from upsnet_amd.operators.modules.mask_roi import MaskROI
    config.dataset.num_classes = C
    rng = np.random.default_rng(N + C)
    H, W = 512, 1024
    rois, delta, prob = _rcnn_inputs(rng, N, C, H, W, peaky)
    im_info = np.array([[H, W, 1.0]], np.float32)
    rs, rb, rc = oops.mask_roi(rois, delta, prob, im_info, C, 0.5, thresh, 100, agn)
    m = MaskROI(True, False, 100, C, nms_thresh=0.5, class_agnostic=agn, score_thresh=thresh)
    s, b, c = m(cu(rois), cu(delta), cu(prob), im_info)
    assert np.array_equal(c.cpu().numpy(), rc)
    assert np.array_equal(s.cpu().numpy(), rs)
    assert np.array_equal(b.cpu().numpy(), rb)
    # clip_boxes=False (modules/mask_roi.py:53-54 skipped): unclipped boxes through the NMS and into the output
    rs, rb, rc = oops.mask_roi(rois, delta, prob, im_info, C, 0.5, thresh, 100, agn, clip=False)
    s, b, c = MaskROI(False, False, 100, C, nms_thresh=0.5, class_agnostic=agn, score_thresh=thresh)(cu(rois), cu(delta), cu(prob), im_info)
    assert np.array_equal(c.cpu().numpy(), rc) and np.array_equal(s.cpu().numpy(), rs) and np.array_equal(b.cpu().numpy(), rb)
    config.dataset.num_classes = 9


# ------------------------------------------------------------------ panoptic head
def _pan_inputs(rng, m, S, C, H, W, ms=28):
    fcn = rng.normal(0, 3, size=(1, S, H, W)).astype(np.float32)
    rois = gen_rois(rng, m, H, W, 8, min(H, W))
    rois[:, 1:] += rng.uniform(-0.9, 0.9, size=(m, 4)).astype(np.float32)
    rois[:, 1:] = np.maximum(rois[:, 1:], 0)
    prob = rng.uniform(0.6, 1.0, size=m).astype(np.float32)
    if m > 3:
        prob[m // 2] = prob[0]
    logit = rng.normal(0.3, 2, size=(m, 1, ms, ms)).astype(np.float32)
    cls = rng.integers(1, C, size=m).astype(np.int64)
    return fcn, rois, prob, logit, cls


@pytest.mark.parametrize("m,S,C,H,W", [(30, 19, 9, 128, 256), (100, 19, 9, 256, 512), (1, 19, 9, 64, 64), (12, 133, 81, 96, 100), (40, 19, 9, 1024, 2048)])
@pytest.mark.parametrize("void", [True, False])
def test_panoptic_head_bitexact(U, m, S, C, H, W, void):
    from upsnet_amd.config.config import config
    config.dataset.num_classes, config.dataset.num_seg_classes = C, S
    rng = np.random.default_rng(m + S)
    fcn, rois, prob, logit, cls = _pan_inputs(rng, m, S, C, H, W)
    ref = oops.panoptic_head(fcn, rois, prob, logit, cls, S, C, enable_void=void)
    keep, num, real = U.mask_removal(cu(rois[:, 1:]), cu(prob), cu(logit), cu(cls), C - 1, (H, W))
    k = int(num.item())
    assert np.array_equal(keep[:k].cpu().numpy(), ref['keep_inds'])
    cmap = cu(oops.class_mapping(S, C))
    pan, sem = U.panoptic_fuse(cu(fcn), S - (C - 1), cu(rois), cu(logit), cu(cls), keep, num, real, cmap, void)
    assert np.array_equal(sem.cpu().numpy()[0], ref['sem'])
    assert np.array_equal(pan.cpu().numpy()[0], ref['panoptic'])
    # materialising module-level path gives the same label map
    energy = U.mask_paste(cu(rois[:, 1:]), cu(logit), keep, num, real, k, (H, W))
    kk = keep[:k]
    seg_inst = U.seg_term(cu(fcn), (cu(rois)[kk] * 4.0)[:, 1:] * 0.25, cu(cls)[kk], cmap)
    pan2 = U.panoptic_argmax(cu(fcn), S - (C - 1), seg_inst, energy, void)
    assert np.array_equal(pan2.cpu().numpy()[0], ref['panoptic'])
    config.dataset.num_classes, config.dataset.num_seg_classes = 9, 19


def test_mask_removal_modules_and_dummy(U):
    from upsnet_amd.operators.modules.mask_removal import MaskRemoval
    from upsnet_amd.operators.modules.unary_logits import SegTerm
    rng = np.random.default_rng(77)
    fcn, rois, prob, logit, cls = _pan_inputs(rng, 20, 19, 9, 96, 160)
    rk, re = oops.mask_removal(rois[:, 1:], prob, logit, cls, (96, 160))
    k, e = MaskRemoval(0.3)(cu(rois[:, 1:]), cu(prob), cu(logit), cu(cls), (96, 160))
    assert np.array_equal(k.cpu().numpy(), rk)
    assert np.array_equal(e.cpu().numpy(), re)
    seg, inst = SegTerm(19)(cu(cls)[k], cu(fcn), cu(rois)[k] * 4.0)
    rseg, rinst = oops.seg_term(cls[rk], fcn, rois[rk] * np.float32(4.0), 19, 9)
    assert np.array_equal(inst.cpu().numpy(), rinst) and np.array_equal(seg.cpu().numpy(), rseg)
    # dummy detection (MaskROI's empty result): keep = [0], one all-zero plane
    k, e = MaskRemoval(0.3)(torch.zeros(1, 4).cuda(), torch.ones(1).cuda(), cu(logit[:1]), torch.zeros(1, dtype=torch.int64).cuda(), (96, 160))
    assert k.tolist() == [0] and e.shape == (1, 1, 96, 160) and not e.any()
    # nothing survives (all logits negative) -> same fallback
    k, e = MaskRemoval(0.3)(cu(rois[:, 1:]), cu(prob), cu(-np.abs(logit) - 1), cu(cls), (96, 160))
    assert k.tolist() == [0] and not e.any()


@pytest.mark.parametrize("m,S,C,Hs,Ws,nhwc", [(30, 19, 9, 32, 64, True), (60, 19, 9, 64, 128, False), (1, 19, 9, 16, 16, True),
                                              (12, 133, 81, 24, 25, True), (40, 19, 9, 256, 512, True)])
def test_panoptic_fuse_with_fused_upsample_bitexact(U, m, S, C, Hs, Ws, nhwc):
    """x4 bilinear upsampling of fcn_score fused into the panoptic kernel == oracle (restated F.interpolate + head);
    agreement with torch's materialised F.interpolate path is reported (ulp-level logit differences only)."""
    from upsnet_amd.config.config import config
    config.dataset.num_classes, config.dataset.num_seg_classes = C, S
    rng = np.random.default_rng(m + S + Hs)
    H, W = Hs * 4, Ws * 4
    _, rois, prob, logit, cls = _pan_inputs(rng, m, S, C, H, W)
    score = rng.normal(0, 3, size=(1, S, Hs, Ws)).astype(np.float32)
    fcn = oracle.upsample_bilinear(score[0], 4)[None]
    ref = oops.panoptic_head(fcn, rois, prob, logit, cls, S, C, enable_void=True)
    keep, num, real = U.mask_removal(cu(rois[:, 1:]), cu(prob), cu(logit), cu(cls), C - 1, (H, W))
    cmap = cu(oops.class_mapping(S, C))
    sc = cu(score)
    if nhwc:
        sc = sc.contiguous(memory_format=torch.channels_last)
    pan, sem = U.panoptic_fuse_up(sc, 4, S - (C - 1), cu(rois), cu(logit), cu(cls), keep, num, real, cmap)
    assert np.array_equal(sem.cpu().numpy()[0], ref['sem'])
    assert np.array_equal(pan.cpu().numpy()[0], ref['panoptic'])
    # vs the materialised path through torch's own upsampling kernel
    up = torch.nn.functional.interpolate(cu(score), None, 4, mode='bilinear', align_corners=False)
    assert float((up - cu(fcn)).abs().max()) < 1e-5
    pan2, _ = U.panoptic_fuse(up, S - (C - 1), cu(rois), cu(logit), cu(cls), keep, num, real, cmap, True)
    assert float((pan2 != pan).float().mean()) < 1e-4
    config.dataset.num_classes, config.dataset.num_seg_classes = 9, 19


def test_panoptic_tail_pack_and_keep_zero_fill(U):
    """ops.panoptic_tail_pack == clamp + index_select + cat (the ATen sequence it replaces), rows past num_keep read row 0;
    mask_removal leaves the rows of keep_inds past the count at 0 without the caller clearing the buffer."""
    torch.manual_seed(5)
    K = 37
    keep = torch.randint(0, K, (K,), device='cuda', dtype=torch.int64)
    keep[3] = K + 5           # out of range on purpose: clamped like keep.clamp(0, K - 1)
    cls = torch.randint(1, 9, (K,), device='cuda', dtype=torch.int64)
    sc = torch.rand(K, device='cuda')
    nums = [torch.tensor([v], dtype=torch.int32, device='cuda') for v in (11, 7, 0)]
    for nk in (1, 9, K):
        num_keep = torch.tensor([nk], dtype=torch.int32, device='cuda')
        kc, ks, cnt = U.panoptic_tail_pack(keep, num_keep, cls, sc, *nums)
        kz = keep.clone()
        kz[nk:] = 0
        kk = kz.clamp(0, K - 1)
        assert torch.equal(kc, cls.index_select(0, kk)) and torch.equal(ks, sc.index_select(0, kk))
        assert cnt.tolist() == [11, 7, 0, nk]
    # keep_inds of the removal: garbage in the caching allocator's block must not survive past the count
    junk = torch.full((64,), 12345, dtype=torch.int64, device='cuda')
    del junk
    m, ms = 6, 28
    rois = torch.tensor([[10., 10., 60., 60.]] * m, device='cuda') + torch.arange(m, device='cuda').view(-1, 1) * 3
    prob = torch.linspace(0.9, 0.4, m, device='cuda')
    logit = torch.full((m, 1, ms, ms), 4.0, device='cuda')
    cidx = torch.ones(m, dtype=torch.int64, device='cuda')
    keep2, num2, _ = U.mask_removal(rois, prob, logit, cidx, 8, (128, 128))
    n = int(num2.item())
    assert 1 <= n < m and bool((keep2[n:] == 0).all())
