"""GPU: model-level parity of the DENSE trunk. The hand-written convolution path (hipconv dispatch -> packed weights -> fp32
MFMA direct / Winograd / deformable / stem / deconv kernels with fused epilogues, multi-map launches, the commuted score tail,
the NHWC fc6 re-layout) against an independent float64 execution of the graph the reference defines
(oracle/dense_ref.py: plain torch functional calls on the same parameters; upsnet/models/resnet.py:347-356, fpn.py:78-104,
rpn.py:52-57, rcnn.py:79-87,132-146, fcn.py:88-108). Tolerance = north_star's "fp32 logits within 1e-4": rtol = atol = 1e-4.

The per-op tests cannot see a wiring / packing-cache / dispatch error in hipconv.py; this one can (missing #4).
Selection outputs (rois, detections) are taken from the product run: they have their own bit-exact tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def _err(got, ref):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    err = (got - ref).abs()
    worst = float((err / (TOL + TOL * ref.abs())).max()) if err.numel() else 0.0
    return worst, (float(err.max()) if err.numel() else 0.0), (float(ref.abs().max()) if err.numel() else 0.0)


def _run(preset, h, w, seed, strict, report_only=()):
    """strict: names that must be within rtol = atol = 1e-4 (elementwise) of the float64 value -- no escape clause (the r11 one, "as close
    as an fp32 library execution that is itself beyond 0.9 of the bound", is gone: r13 keeps the 3x3 layers upstream of a deformable chain
    on the F(2x2) form instead, models/resnet.py _Block.feeds_deformable); report_only: names that are compared and printed but not asserted. Every other name must be within 1e-4 OR at most 3x as far from the float64 value as a plain fp32
    library execution of the same graph (torch / MIOpen convolutions, oracle.dense_ref in float32) -- since r08 (calibrated synthetic
    statistics, activations O(1-10) as in a trained network) no tensor of either model needs that escape.

    Data-dependent samplers are compared AT IDENTICAL SAMPLING POSITIONS: the float64 execution samples every deformable layer
    (the 2-3 of the semantic head; the 30 bottlenecks of the R101-DCN backbone) at the offsets the product recorded, and the offset
    predictions themselves are compared layer by layer along that same chain, strictly. (Free-running, a deformable layer multiplies
    an offset difference by the local feature gradient, which on the spatially white feature maps of a random-noise image is ~1 per
    pixel: ~3x per layer, so 30 layers in sequence are chaotic for ANY fp32 execution. 'fcn_score' free-running is report_only.)"""
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, config
    from upsnet_amd.models import hipconv
    from oracle import dense_ref
    update_config_dict(preset)
    try:
        from upsnet_amd.synthetic import build_model, make_image
        model = build_model()
        data = make_image(h, w, seed=seed, device='cuda')
        model.taps = {}
        sub = model.fcn_head.fcn_subnet
        sub.taps = {}
        hipconv.TRACE = []
        try:
            with torch.no_grad():
                out = model(data)
            trace = hipconv.TRACE
        finally:
            hipconv.TRACE = None
        t, model.taps = model.taps, None
        offs, sub.taps = sub.taps['offsets'], None
        bb_offs = [r['offsets'][0] for r in trace if r['kind'] == 'dcn' and r['form'] == 'dcn_fused']   # backbone bottlenecks, graph order
        del trace
        n = int(t['n_rois'].item())
        ms = config.network.mask_size
        args = (model, data, t['rois'][:n], t['det_boxes'], t['pan_boxes'], ms)
        ref = dense_ref.dense_reference(*args, fcn_offsets=offs, backbone_offsets=bb_offs or None)
        lib = dense_ref.dense_reference(*args, dtype=torch.float32, fcn_offsets=offs, backbone_offsets=bb_offs or None)

        def pan_logit(r):
            return r['mask_logit_pan'].gather(1, t['pan_cls'].view(-1, 1, 1, 1).expand(-1, -1, ms, ms).to(r['mask_logit_pan'].device))
        pairs = {}
        for l in range(5):
            pairs['rpn_cls_prob_p%d' % (l + 2)] = (t['rpn_cls_prob'][l], ref['rpn_cls_prob'][l], lib['rpn_cls_prob'][l])
            pairs['rpn_bbox_pred_p%d' % (l + 2)] = (t['rpn_bbox_pred'][l], ref['rpn_bbox_pred'][l], lib['rpn_bbox_pred'][l])
        pairs['cls_prob'] = (t['cls_prob'][:n], ref['cls_prob'], lib['cls_prob'])
        pairs['bbox_pred'] = (t['bbox_pred'][:n], ref['bbox_pred'], lib['bbox_pred'])
        pairs['mask_probs'] = (out['mask_probs'], torch.sigmoid(ref['mask_logit_det']), torch.sigmoid(lib['mask_logit_det']))
        pairs['pan_mask_logit'] = (t['pan_logit'], pan_logit(ref), pan_logit(lib))
        pairs['fcn_score'] = (t['fcn_score'], ref['fcn_score'], lib['fcn_score'])
        pairs['fcn_score_at_recorded_offsets'] = (t['fcn_score'], ref['fcn_score_given'], lib['fcn_score_given'])
        for i, per_level in enumerate(offs):
            for l, o in enumerate(per_level):
                pairs['fcn_offset_layer%d_p%d' % (i, l + 2)] = (o, ref['fcn_offsets'][i][l], lib['fcn_offsets'][i][l])
        for i, o in enumerate(bb_offs):
            pairs['backbone_offset_%02d' % i] = (o, ref['backbone_offsets'][i], lib['backbone_offsets'][i])
        rep, bad = {}, {}
        for name, (got, r64, r32) in pairs.items():
            worst, max_abs, max_ref = _err(got, r64)
            ent = dict(worst_over_bound=round(worst, 3), max_abs=max_abs, max_ref=max_ref)
            ok = worst <= 1.0
            lib_worst, lib_abs, _ = _err(r32, r64)
            ent.update(fp32_library_worst_over_bound=round(lib_worst, 3), fp32_library_max_abs=lib_abs)
            if not ok and name not in strict and name not in report_only:
                ok = max_abs <= 3.0 * lib_abs
                ent['criterion'] = '<= 3x the fp32 library execution'
            rep[name] = ent
            if not ok and name not in report_only:
                bad[name] = ent
        assert not bad, bad
        assert n > 50 and t['det_boxes'].shape[0] >= 1 and t['pan_boxes'].shape[0] >= 1
        assert not any('criterion' in v for v in rep.values()), {k: v for k, v in rep.items() if 'criterion' in v}
        return rep
    finally:
        update_config_dict(CITYSCAPES_R50)


_RPN = ['rpn_cls_prob_p%d' % l for l in range(2, 7)] + ['rpn_bbox_pred_p%d' % l for l in range(2, 7)]
_STRICT = _RPN + ['cls_prob', 'bbox_pred', 'mask_probs', 'pan_mask_logit', 'fcn_score_at_recorded_offsets']


def _summary(rep):
    keep = {k: v for k, v in rep.items() if not k.startswith(('backbone_offset_', 'fcn_offset_'))}
    offs = [v['worst_over_bound'] for k, v in rep.items() if k.startswith(('backbone_offset_', 'fcn_offset_'))]
    print({k: (v['worst_over_bound'], v.get('fp32_library_worst_over_bound')) for k, v in keep.items()}, 'offset predictions: worst', max(offs))


@pytest.mark.parametrize("h,w", [(256, 512), (1024, 2048)])
def test_trunk_logits_vs_fp64_reference_upsnet50(h, w):
    """UPSNet-50: RPN / box / mask heads, the panoptic mask logits and the semantic head (at the recorded offsets) strictly within 1e-4
    at both sizes, 1024x2048 = the benchmark configuration included; every offset prediction strictly."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    rep = _run(CITYSCAPES_R50, h, w, seed=3, strict=_STRICT + ['fcn_offset_layer%d_p%d' % (i, l) for i in range(2) for l in range(2, 6)],
               report_only=['fcn_score'])
    _summary(rep)


@pytest.mark.parametrize("h,w", [(200, 333), (800, 1333), (1024, 2048)])
def test_trunk_logits_vs_fp64_reference_upsnet101_dcn(h, w):
    """(1024x2048: the Cityscapes-shaped half of BASELINE configs[4]'s mixed stream, VERDICT r03 next #1b.) BASELINE configs[3]: R101 with DCN v1 in res3-res5 (30 deformable layers in front of everything), GAP in the FPN, 3 FCN
    layers, 81 / 133 classes, 300 proposals: the same strict list, every deformable layer sampled at the product's recorded offsets
    and every offset prediction (30 backbone + 3 x 4 head) strictly within 1e-4 of the float64 prediction along that chain."""
    from upsnet_amd.config.config import COCO_R101_DCN
    rep = _run(COCO_R101_DCN, h, w, seed=4, strict=_STRICT + ['backbone_offset_%02d' % i for i in range(30)] +
               ['fcn_offset_layer%d_p%d' % (i, l) for i in range(3) for l in range(2, 6)], report_only=['fcn_score'])
    _summary(rep)
