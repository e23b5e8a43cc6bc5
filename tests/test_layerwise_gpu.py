"""GPU: "fp32 logits within 1e-4 -- per op, on identical inputs" (north_star / BASELINE.md section 2) for EVERY convolution launch of
the model at its REAL size, as models/hipconv.py dispatches it.

One eager product forward runs with hipconv.TRACE set: every convolution launch (1x1 GEMM kernel, res2 pair kernel, Winograd 64- / 32-tile
and split-K forms, implicit-GEMM incl. split-K, multi-map launches, concatenated RPN heads, 7x7 stem, 2x2 deconvolution, fused deformable
convolutions) records its module, its input, the operands of its fused epilogue (bias = folded frozen BN, residual, nearest x2
upsampled residual, ReLU) and its output. Each record is then replayed in float64 with plain torch calls on the SAME input tensor
(reference graph: upsnet/models/resnet.py:53-100,155-175, fpn.py:78-104, rpn.py:52-57, rcnn.py:79-87, fcn.py:29-58,88-108; deformable
sampling deform_conv_kernel.cu:88-118,227-240 via oracle.dense_ref.deform_conv at the recorded offsets) and compared at
rtol = atol = 1e-4, strictly: no library-relative escape. At 1024x2048 this covers each of the C1 convolution shapes at the size and in
the kernel form the benchmark runs it.
"""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _replay(rec):
    """float64 value of one recorded launch from its recorded input."""
    from oracle import dense_ref
    m, D = rec['module'], torch.float64
    w = m.weight.detach().to(D)
    b = None if m.bias is None else m.bias.detach().to(D)
    if rec['kind'] == 'conv':
        y = F.conv2d(rec['x'].to(D), w, b, m.stride, m.padding, m.dilation, m.groups)
        if rec['residual'] is not None:
            r = rec['residual'].to(D)
            y = y + (F.interpolate(r, scale_factor=2, mode='nearest') if rec['residual_up'] else r)
        return [F.relu(y) if rec['relu'] else y], [rec['out']]
    if rec['kind'] == 'stem_pool':     # stem convolution + ReLU + 3x3 / 2 max-pool in one launch (resnet.py:347-356)
        y = F.relu(F.conv2d(rec['x'].to(D), w, b, m.stride, m.padding))
        return [F.max_pool2d(y, 3, stride=2, padding=1)], [rec['out']]
    if rec['kind'] == 'deconv':
        y = F.conv_transpose2d(rec['x'].to(D), w, b, m.stride, m.padding)
        return [F.relu(y) if rec['relu'] else y], [rec['out']]
    assert rec['kind'] == 'dcn'
    saved, dense_ref.D = dense_ref.D, D
    try:
        refs = [dense_ref.deform_conv(x.to(D), o.to(D), m, relu=rec['relu']) for x, o in zip(rec['xs'], rec['offsets'])]
    finally:
        dense_ref.D = saved
    return refs, list(rec['outs'])


def _describe(rec):
    m = rec['module']
    x = rec['x'] if 'x' in rec else rec['xs'][0]
    return "%s %s k%s s%s in %s [%s]" % (rec['kind'], '%d->%d' % (m.in_channels, m.out_channels), tuple(m.kernel_size), tuple(m.stride),
                                         tuple(x.shape), rec['form'])


def _check_model(preset, h, w, seed, need_forms, min_shapes, precision='fp32', wino36=True):
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    from upsnet_amd.models import hipconv
    update_config_dict(preset)
    saved_precision, hipconv.PRECISION = hipconv.PRECISION, precision
    saved_w36, hipconv.WINO36 = hipconv.WINO36, wino36
    try:
        from upsnet_amd.synthetic import build_model, make_image
        model = build_model()
        data = make_image(h, w, seed=seed, device='cuda')
        hipconv.TRACE, n_fallbacks = [], len(hipconv.FALLBACKS)
        try:
            with torch.no_grad():
                out = model(data)
            torch.cuda.synchronize()
            trace = hipconv.TRACE
        finally:
            hipconv.TRACE = None
        assert out['cls_inds'].numel() >= 1
        bad, forms, shapes, worst_all = [], set(), set(), 0.0
        with torch.no_grad():
            for rec in trace:
                refs, gots = _replay(rec)
                for ref, got in zip(refs, gots):
                    assert ref.shape == got.shape, (_describe(rec), ref.shape, got.shape)
                    err = (got.double() - ref).abs()
                    worst = float((err / (TOL + TOL * ref.abs())).max()) if err.numel() else 0.0
                    worst_all = max(worst_all, worst)
                    if not worst <= 1.0:
                        bad.append((_describe(rec), round(worst, 3), float(err.max()), float(ref.abs().max())))
                forms.add(rec['form'])
                m = rec['module']
                x = rec['x'] if 'x' in rec else rec['xs'][0]
                shapes.add((rec['kind'], m.in_channels, m.out_channels, tuple(m.kernel_size), tuple(m.stride), tuple(x.shape[2:])))
                del refs
        assert not bad, bad
        missing = [f for f in need_forms if not any(g.startswith(f) for g in forms)]
        assert not missing, (missing, sorted(forms))
        assert len(shapes) >= min_shapes, (len(shapes), min_shapes)
        assert len(hipconv.FALLBACKS) == n_fallbacks, hipconv.FALLBACKS[n_fallbacks:]
        print("%s%d launches, %d distinct layer shapes, forms %s, worst error / bound %.3f" %
              ('' if precision == 'fp32' else precision + ': ', len(trace), len(shapes), sorted(forms), worst_all))
    finally:
        hipconv.WINO36 = saved_w36
        hipconv.PRECISION = saved_precision
        update_config_dict(CITYSCAPES_R50)


_C1_FORMS = ['stem', 'conv1x1', 'pair(conv3)', 'pair(conv1)', 'winograd36', 'winograd36 multi', 'winograd36 roi', 'winograd tm32', 'winograd splitk', 'igemm', 'conv1x1 ksw',
             'igemm multi cat', 'deconv2x2', 'dcn_fused multi']


@pytest.mark.parametrize("h,w", [(256, 512), (1024, 2048)])
def test_every_convolution_launch_vs_fp64_on_identical_inputs_upsnet50(h, w):
    """C1 (UPSNet-50 Cityscapes). At 1024x2048: each of its convolution shapes at the real size, in the kernel form hipconv picks there
    (F(4x4) Winograd on FPN-P2 / the RPN launch, split-K on res5, the small-tile 1x1 kernel on the P5 lateral, the res2 pair kernel, ...)."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    full = h * w >= 1 << 21
    _check_model(CITYSCAPES_R50, h, w, seed=3, need_forms=_C1_FORMS if full else ['stem', 'deconv2x2', 'dcn_fused multi'],
                 min_shapes=35 if full else 30)


def test_every_convolution_launch_vs_fp64_upsnet50_without_f4x4():
    """UPSNET_WINO36=0: the largest 3x3 layers (FPN P2, the RPN launch) on the 64-tile F(2x2,3x3) form they ran on up to round 4."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    _check_model(CITYSCAPES_R50, 1024, 2048, seed=3, need_forms=['winograd tm64', 'winograd tm64 multi', 'winograd tm32', 'winograd tm32 + tail tn32'], min_shapes=35, wino36=False)


@pytest.mark.parametrize("h,w", [(200, 333), (800, 1333)])
def test_every_convolution_launch_vs_fp64_on_identical_inputs_upsnet101_dcn(h, w):
    """C2 (BASELINE configs[3]): R101 with 30 deformable bottlenecks (each checked at its recorded offsets = identical sampling
    positions, its offset prediction as a convolution of its own), GAP, 3 FCN layers, 81 / 133 classes."""
    from upsnet_amd.config.config import COCO_R101_DCN
    # (r13) at 800x1333 the small-tile kernels carry the bottlenecks' conv1 layers / the P5 lateral (conv1x1 ksw) and the offset predictors of res4 / res5
    # (conv3x3 ksw): each of those launches is replayed in float64 like every other one
    _check_model(COCO_R101_DCN, h, w, seed=4, need_forms=['stem', 'deconv2x2', 'dcn_fused'] + (['conv1x1 ksw', 'conv3x3 ksw'] if h >= 800 else []), min_shapes=30)


# ----------------------------------------------------------------------------- bf16 mode (BASELINE.json configs[2]) and configs[4]
def _rb(t):
    """round to nearest even to bf16, as float64 (what the bf16 kernels do to an fp32 operand on the way in / when they pack weights)."""
    return t.detach().to(torch.bfloat16).double()


def _replay_bf16(rec):
    """float64 value of one recorded launch of the bf16 mode ON THE bf16-ROUNDED OPERANDS, + the bound that launch must meet:
      'f32'   |err| <= 1e-4 + 1e-4 |ref|                       launch on an fp32 kernel, or bf16 products (exact in fp32) accumulated in
                                                               fp32 with an fp32 result
      'b16'   |err| <= 2^-8 |ref| + 1e-4 + 1e-4 |ref|          the same, the fp32 result rounded ONCE to a bf16 output
      'dcn16' |err| <= 1e-4 + 1e-4 |ref| + 3 * 2^-8 smax wmax  deformable: the blended sample (fp32 expression order of
                                                               deform_conv_kernel.cu:88-118) is rounded to bf16 before the product; a sample
                                                               within fp32 rounding of a bf16 midpoint may round the other way than the float64
                                                               blend does -- up to 3 such one-step flips (2^-8 x largest sample x largest
                                                               weight each) per output element are allowed
      'chain' fused kernels with INTERNAL bf16 roundings (the one-launch bottlenecks: t1, t2 rounded where the separate launches round
              them): max |err| <= 0.02 scale, mean |err| <= 2e-3 scale, >= 99 % of the elements within one bf16 step of the float64
              result (tests/test_bottleneck_bf16_gpu.py states the same bound for the isolated kernel)."""
    from oracle import dense_ref
    m, D = rec['module'], torch.float64
    kind, form = rec['kind'], rec['form']
    if kind == 'block':
        blk = m
        st = blk.conv1.stride[0]
        x = rec['x'].double()
        c = lambda mod, t, **kw: F.conv2d(t, _rb(mod.weight), mod.bias.detach().double(), **kw)
        t1 = _rb(F.relu(c(blk.conv1, x, stride=st)))
        t2 = _rb(F.relu(c(blk.conv2, t1, padding=1)))
        sc = x if blk.downsample is None else c(blk.downsample[0], x, stride=st)
        return [(F.relu(c(blk.conv3, t2) + sc), rec['out'], 'chain', None)]
    bf = form.startswith('bf16') or form.endswith('bf16')
    if not bf:
        refs, gots = _replay(rec)
        return [(r, g, 'f32', None) for r, g in zip(refs, gots)]
    w = _rb(m.weight)
    b = None if m.bias is None else m.bias.detach().to(D)
    if kind == 'conv':
        y = F.conv2d(_rb(rec['x']), w, b, m.stride, m.padding, m.dilation, m.groups)
        if rec['residual'] is not None:
            r = rec['residual'].to(D)
            y = y + (F.interpolate(r, scale_factor=2, mode='nearest') if rec['residual_up'] else r)
        y = F.relu(y) if rec['relu'] else y
        return [(y, rec['out'], 'b16' if rec['out'].dtype == torch.bfloat16 else 'f32', None)]
    if kind == 'stem_pool':
        y = F.max_pool2d(F.relu(F.conv2d(_rb(rec['x']), w, b, m.stride, m.padding)), 3, stride=2, padding=1)
        return [(y, rec['out'], 'b16', None)]
    if kind == 'deconv':
        y = F.conv_transpose2d(_rb(rec['x']), w, b, m.stride, m.padding)
        y = F.relu(y) if rec['relu'] else y
        return [(y, rec['out'], 'b16' if rec['out'].dtype == torch.bfloat16 else 'f32', None)]
    assert kind == 'dcn'
    res = []
    k = m.kernel_size[0]
    saved, dense_ref.D = dense_ref.D, D
    try:
        for x, o, got in zip(rec['xs'], rec['offsets'], rec['outs']):
            out, smax = None, 0.0
            C = x.shape[1]
            for c0 in range(0, C, 64):
                col = dense_ref.deform_im2col(x[:, c0:c0 + 64].to(D), o.to(D), None, k, m.padding[0], m.stride[0], m.dilation[0], 1)
                smax = max(smax, float(col.abs().max()))
                part = torch.einsum('ock,bckhw->bohw', w[:, c0:c0 + 64].reshape(w.shape[0], -1, k * k), _rb(col))
                out = part if out is None else out + part
                del col
            if b is not None:
                out = out + b.view(1, -1, 1, 1)
            res.append((F.relu(out) if rec['relu'] else out, got, 'dcn16', 3.0 * 2.0 ** -8 * smax * float(w.abs().max())))
    finally:
        dense_ref.D = saved
    return res


def _judge(ref, got, bound, extra):
    """-> (worst error / bound, description of a violation or None)."""
    err = (got.double() - ref).abs()
    if not err.numel():
        return 0.0, None
    if bound == 'chain':
        scale = float(ref.abs().max())
        near = float((err <= ref.abs() * 2.0 ** -7 + 1e-2 * scale * 2.0 ** -4).double().mean())
        worst = max(float(err.max()) / (0.02 * scale), float(err.mean()) / (2e-3 * scale))
        ok = worst <= 1.0 and near > 0.99
        return worst, None if ok else 'chain: max %.3g mean %.3g scale %.3g near %.4f' % (float(err.max()), float(err.mean()), scale, near)
    lim = TOL + TOL * ref.abs()
    if bound == 'b16':
        lim = lim + 2.0 ** -8 * ref.abs()
    elif bound == 'dcn16':
        lim = lim + extra
    worst = float((err / lim).max())
    return worst, None if worst <= 1.0 else '%s: worst %.3f x bound, max err %.3g, max |ref| %.3g' % (bound, worst, float(err.max()), float(ref.abs().max()))


def _check_model_bf16(preset, h, w, seed, need_forms):
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    from upsnet_amd.models import hipconv
    update_config_dict(preset)
    saved = hipconv.PRECISION
    try:
        from upsnet_amd.synthetic import build_model, make_image
        hipconv.PRECISION = 'bf16'
        model = build_model()
        data = make_image(h, w, seed=seed, device='cuda')
        hipconv.TRACE, n_fallbacks = [], len(hipconv.FALLBACKS)
        try:
            with torch.no_grad():
                out = model(data)
            torch.cuda.synchronize()
            trace = hipconv.TRACE
        finally:
            hipconv.TRACE = None
        assert out['cls_inds'].numel() >= 1
        bad, forms, worst_by = [], {}, {}
        with torch.no_grad():
            for rec in trace:
                for ref, got, bound, extra in _replay_bf16(rec):
                    assert ref.shape == got.shape, (rec['form'], ref.shape, got.shape)
                    worst, why = _judge(ref, got, bound, extra)
                    worst_by[bound] = max(worst_by.get(bound, 0.0), worst)
                    if why is not None:
                        m = rec['module']
                        bad.append((rec['kind'], rec['form'], type(m).__name__, tuple(got.shape), why))
                    del ref
                forms[rec['form']] = forms.get(rec['form'], 0) + 1
        print("bf16 mode %dx%d: %d launches, forms %s, worst error / bound per class %s" %
              (h, w, len(trace), dict(sorted(forms.items())), {k: round(v, 3) for k, v in sorted(worst_by.items())}))
        assert not bad, bad
        missing = [f for f in need_forms if not any(g.startswith(f) for g in forms)]
        assert not missing, (missing, sorted(forms))
        assert len(hipconv.FALLBACKS) == n_fallbacks, hipconv.FALLBACKS[n_fallbacks:]
    finally:
        hipconv.PRECISION = saved
        update_config_dict(CITYSCAPES_R50)


# the kernel instances BASELINE configs[2] is benchmarked on at 1024x2048 (profiles/r09_bf16_kernel_stats.txt): one-launch identity and
# projection bottlenecks, 3x3 layers on 8-row / 256-channel and on 2-row / 128-channel weights-from-L2 tiles, the no-LDS 1x1 kernel,
# the general bf16 GEMM kernel, fused stem + pool, bf16 transposed convolution, bf16 fused deformable convolution
_C1_BF16_FORMS = ['stem + pool bf16', 'bottleneck_bf16', 'bottleneck_proj_bf16', 'bf16 conv3x3_wreg<8,2>', 'bf16 conv3x3_wreg<2,1>',
                  'bf16 conv1x1_wreg<4,', 'bf16 conv_bf16<1,', 'deconv2x2 bf16', 'dcn_fused multi bf16']


@pytest.mark.parametrize("h,w", [(256, 512), (1024, 2048)])
def test_every_launch_of_the_bf16_mode_vs_fp64_on_rounded_operands_upsnet50(h, w):
    """BASELINE configs[2] (VERDICT r03 next #1a): every convolution launch of UPSNet-50 in the bf16 mode -- fused bottlenecks,
    conv3x3_wreg / conv1x1_wreg / conv_bf16, stem + pool, bf16 deconvolution, bf16 fused DCN, and the layers that stay on the fp32
    kernels -- against float64 on the bf16-rounded operands, with the per-launch bounds of _replay_bf16; at 1024x2048 the kernel
    instances must be the ones the benchmark runs on."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    full = h * w >= 1 << 21
    _check_model_bf16(CITYSCAPES_R50, h, w, seed=3, need_forms=_C1_BF16_FORMS if full else ['stem + pool bf16', 'bf16 conv3x3_wreg<8,2>', 'bf16 conv1x1_wreg<4,', 'dcn_fused multi bf16'])


def test_every_convolution_launch_vs_fp64_upsnet101_dcn_at_1024x2048():
    """BASELINE configs[4], its Cityscapes-shaped half (VERDICT r03 next #1b): UPSNet-101-DCN at 1024x2048 -- 30 deformable
    bottlenecks at their recorded offsets, every launch strictly at 1e-4 against float64, in the forms hipconv picks at that size."""
    from upsnet_amd.config.config import COCO_R101_DCN
    _check_model(COCO_R101_DCN, 1024, 2048, seed=6, need_forms=['stem', 'deconv2x2', 'dcn_fused', 'dcn_fused multi', 'winograd36', 'conv1x1',
                                                                'pair(conv3)'], min_shapes=40)


def test_every_launch_of_the_bf16x3_mode_vs_fp64_at_1024x2048():
    """The fp32-equivalent three-term split on the bf16 matrix cores (`--conv-precision bf16x3`, 168-181 images/s quoted) at the size it is
    quoted for (VERDICT r04 weak #1b: it was checked at 256x512 only): every launch strictly at rtol = atol = 1e-4 against float64 on
    the UNROUNDED operands -- the same bar as the fp32 kernels -- and the split kernels really ran."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    _check_model(CITYSCAPES_R50, 1024, 2048, seed=3, need_forms=['bf16x3 ', 'stem', 'deconv2x2', 'dcn_fused multi'], min_shapes=35, precision='bf16x3')


@pytest.mark.parametrize("h,w", [(800, 1333)])
def test_every_launch_of_the_bf16_mode_vs_fp64_on_rounded_operands_upsnet101_dcn(h, w):
    """UPSNet-101-DCN in the bf16 mode at the size `profiles/*_bench_c3_bf16.log` quotes (VERDICT r04 missing #3): 30 `dcn_fused bf16`
    bottlenecks (resnet.py:102-153) each at its recorded offsets on the rounded operands with the 'dcn16' bound, their fp32 offset
    predictors, the one-launch bf16 bottlenecks of res2, every other launch -- with the per-launch bounds of _replay_bf16."""
    from upsnet_amd.config.config import COCO_R101_DCN
    _check_model_bf16(COCO_R101_DCN, h, w, seed=4, need_forms=['stem + pool bf16', 'bottleneck_bf16', 'dcn_fused bf16', 'dcn_fused multi bf16',
                                                               'deconv2x2 bf16', 'bf16 conv'])
