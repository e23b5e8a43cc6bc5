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


def _run(preset, h, w, seed, strict, offsets_chain=True):
    """strict: names that must be within rtol = atol = 1e-4 (elementwise) of the float64 value. The others must be within 1e-4 OR
    at most 3x as far from the float64 value as a plain fp32 library execution of the same graph (torch / MIOpen convolutions,
    oracle.dense_ref in float32), i.e. as accurate as the reference's own fp32 path. Two things make the elementwise 1e-4
    unattainable for ANY fp32 execution on these synthetic weights (frozen identity BN: activations of magnitude 50-140):
    tensors behind deformable layers (an offset difference of 1e-5 px moves a sample of magnitude-100 features by 1e-3), and, at
    1024x2048, near-zero elements of tensors whose scale is ~50 (the library execution itself is 2.4x over the bound there)."""
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, config
    from oracle import dense_ref
    update_config_dict(preset)
    try:
        from upsnet_amd.synthetic import build_model, make_image
        model = build_model(cls_gain=0.3)
        data = make_image(h, w, seed=seed, device='cuda')
        model.taps = {}
        sub = model.fcn_head.fcn_subnet
        sub.taps = {}
        with torch.no_grad():
            out = model(data)
        t, model.taps = model.taps, None
        offs, sub.taps = sub.taps['offsets'], None
        n = int(t['n_rois'].item())
        ms = config.network.mask_size
        args = (model, data, t['rois'][:n], t['det_boxes'], t['pan_boxes'], ms)
        ref = dense_ref.dense_reference(*args, fcn_offsets=offs)
        lib = dense_ref.dense_reference(*args, dtype=torch.float32, fcn_offsets=offs)

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
        # the semantic head at IDENTICAL sampling positions (the product's recorded offsets): strict; and the offset predictions
        # themselves, layer by layer, against the float64 prediction along that same chain: strict
        # (only meaningful when the features in front of the head are themselves strictly equal: not behind a DCN backbone)
        if offsets_chain:
            pairs['fcn_score_at_recorded_offsets'] = (t['fcn_score'], ref['fcn_score_given'], lib['fcn_score_given'])
            for i, per_level in enumerate(offs):
                for l, o in enumerate(per_level):
                    pairs['fcn_offset_layer%d_p%d' % (i, l + 2)] = (o, ref['fcn_offsets'][i][l], lib['fcn_offsets'][i][l])
        rep, bad = {}, {}
        for name, (got, r64, r32) in pairs.items():
            worst, max_abs, max_ref = _err(got, r64)
            ent = dict(worst_over_bound=round(worst, 3), max_abs=max_abs, max_ref=max_ref)
            ok = worst <= 1.0
            if r32 is not None:
                lib_worst, lib_abs, _ = _err(r32, r64)
                ent.update(fp32_library_worst_over_bound=round(lib_worst, 3), fp32_library_max_abs=lib_abs)
                if not ok and name not in strict:
                    ok = max_abs <= 3.0 * lib_abs
                    ent['criterion'] = '<= 3x the fp32 library execution'
            rep[name] = ent
            if not ok:
                bad[name] = ent
        assert not bad, bad
        assert n > 50 and t['det_boxes'].shape[0] >= 1 and t['pan_boxes'].shape[0] >= 1
        return rep
    finally:
        update_config_dict(CITYSCAPES_R50)


_RPN = ['rpn_cls_prob_p%d' % l for l in range(2, 7)] + ['rpn_bbox_pred_p%d' % l for l in range(2, 7)]


@pytest.mark.parametrize("h,w", [(256, 512), (1024, 2048)])
def test_trunk_logits_vs_fp64_reference_upsnet50(h, w):
    """UPSNet-50 has no deformable layer in front of the RPN / box / mask heads: all of them strictly within 1e-4; the semantic
    head strictly at the recorded offsets."""
    from upsnet_amd.config.config import CITYSCAPES_R50
    strict = _RPN + ['cls_prob', 'bbox_pred', 'mask_probs'] + (['pan_mask_logit', 'fcn_score_at_recorded_offsets'] if h * w < 1 << 20 else [])
    rep = _run(CITYSCAPES_R50, h, w, seed=3, strict=strict)
    print({k: (v['worst_over_bound'], v.get('fp32_library_worst_over_bound')) for k, v in rep.items()})


@pytest.mark.parametrize("h,w", [(200, 333), (800, 1333)])
def test_trunk_logits_vs_fp64_reference_upsnet101_dcn(h, w):
    """BASELINE configs[3]: R101 with DCN v1 in res3-res5 (30 deformable layers in front of everything), GAP in the FPN, 3 FCN
    layers, 81 / 133 classes, 300 proposals."""
    from upsnet_amd.config.config import COCO_R101_DCN
    rep = _run(COCO_R101_DCN, h, w, seed=4, strict=[], offsets_chain=False)
    print({k: (v['worst_over_bound'], v.get('fp32_library_worst_over_bound')) for k, v in rep.items()})
