"""End-to-end GPU tests: the fused MI355X pipeline vs the reference-shaped module dataflow, and the
whole forward vs the oracle composite on a small synthetic image."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    from upsnet_amd.synthetic import build_model, make_image
    model = build_model(cls_gain=60.0)
    data = make_image(256, 512, seed=0, device='cuda')
    return model, data


def test_fused_equals_modules(setup):
    model, data = setup
    with torch.no_grad():
        model.pipeline = 'fused'
        a = model(data)
        model.pipeline = 'modules'
        b = model(data)
        model.pipeline = 'fused'
    for k in ('cls_probs', 'pred_boxes', 'cls_inds', 'panoptic_cls_inds', 'panoptic_cls_probs', 'fcn_outputs', 'panoptic_outputs'):
        assert torch.equal(a[k], b[k]), k
    np.testing.assert_allclose(a['mask_probs'].cpu().numpy(), b['mask_probs'].cpu().numpy(), rtol=1e-4, atol=1e-5)


def test_forward_vs_oracle_composite(setup):
    from oracle.forward import forward_oracle
    model, data = setup
    with torch.no_grad():
        out = model(data)
    ref = forward_oracle(model, data)
    # identical conv backend is not available on the CPU: compare through the tolerance-free stages by
    # feeding the oracle the device's conv outputs (forward_oracle(..., taps=...)) -- see oracle/forward.py
    assert ref['n_rois'] > 0
    taps = ref['taps']
    assert np.array_equal(out['pred_boxes'].cpu().numpy(), taps['pred_boxes'])
    assert np.array_equal(out['cls_inds'].cpu().numpy(), taps['cls_inds'])
    assert np.array_equal(out['panoptic_cls_inds'].cpu().numpy(), taps['panoptic_cls_inds'])
    assert np.array_equal(out['fcn_outputs'].cpu().numpy()[0], taps['fcn_outputs'])
    assert np.array_equal(out['panoptic_outputs'].cpu().numpy()[0], taps['panoptic_outputs'])


def test_repeatable(setup):
    model, data = setup
    with torch.no_grad():
        a = model(data)
        b = model(data)
    for k in a:
        assert torch.equal(a[k], b[k]), k
