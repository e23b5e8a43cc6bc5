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


def _close(name, got, ref, report):
    got, ref = got.detach().double().cpu(), ref.detach().double().cpu()
    assert got.shape == ref.shape, (name, got.shape, ref.shape)
    err = (got - ref).abs()
    bound = TOL + TOL * ref.abs()
    worst = float((err / bound).max()) if err.numel() else 0.0
    report[name] = dict(max_abs=float(err.max()) if err.numel() else 0.0, max_ref=float(ref.abs().max()) if err.numel() else 0.0,
                        worst_over_bound=worst)
    return worst <= 1.0


def _run(preset, h, w, seed):
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, config
    from oracle import dense_ref
    update_config_dict(preset)
    try:
        from upsnet_amd.synthetic import build_model, make_image
        model = build_model(cls_gain=0.3)
        data = make_image(h, w, seed=seed, device='cuda')
        model.taps = {}
        with torch.no_grad():
            out = model(data)
        t, model.taps = model.taps, None
        n = int(t['n_rois'].item())
        ref = dense_ref.dense_reference(model, data, t['rois'][:n], t['det_boxes'], t['pan_boxes'], config.network.mask_size)
        rep, ok = {}, []
        for l in range(5):
            ok.append(_close('rpn_cls_prob_p%d' % (l + 2), t['rpn_cls_prob'][l], ref['rpn_cls_prob'][l], rep))
            ok.append(_close('rpn_bbox_pred_p%d' % (l + 2), t['rpn_bbox_pred'][l], ref['rpn_bbox_pred'][l], rep))
        ok.append(_close('fcn_score', t['fcn_score'], ref['fcn_score'], rep))
        ok.append(_close('cls_prob', t['cls_prob'][:n], ref['cls_prob'], rep))
        ok.append(_close('bbox_pred', t['bbox_pred'][:n], ref['bbox_pred'], rep))
        ok.append(_close('mask_probs', out['mask_probs'], torch.sigmoid(ref['mask_logit_det']), rep))
        ms = config.network.mask_size
        pan_ref = ref['mask_logit_pan'].gather(1, t['pan_cls'].view(-1, 1, 1, 1).expand(-1, -1, ms, ms).to(ref['mask_logit_pan'].device))
        ok.append(_close('pan_mask_logit', t['pan_logit'], pan_ref, rep))
        bad = {k: v for k, v in rep.items() if v['worst_over_bound'] > 1.0}
        assert all(ok), bad
        assert n > 50 and t['det_boxes'].shape[0] >= 1 and t['pan_boxes'].shape[0] >= 1
        return rep
    finally:
        update_config_dict(CITYSCAPES_R50)


@pytest.mark.parametrize("h,w", [(256, 512), (1024, 2048)])
def test_trunk_logits_vs_fp64_reference_upsnet50(h, w):
    from upsnet_amd.config.config import CITYSCAPES_R50
    rep = _run(CITYSCAPES_R50, h, w, seed=3)
    print({k: round(v['worst_over_bound'], 3) for k, v in rep.items()})


@pytest.mark.parametrize("h,w", [(200, 333), (800, 1333)])
def test_trunk_logits_vs_fp64_reference_upsnet101_dcn(h, w):
    """BASELINE configs[3]: R101 with DCN v1 in res3-res5, GAP in the FPN, 3 FCN layers, 81 / 133 classes, 300 proposals."""
    from upsnet_amd.config.config import COCO_R101_DCN
    rep = _run(COCO_R101_DCN, h, w, seed=4)
    print({k: round(v['worst_over_bound'], 3) for k, v in rep.items()})
