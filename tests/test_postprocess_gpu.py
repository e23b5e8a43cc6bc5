"""GPU: get_unified_pan_result on the device vs the numpy oracle and vs the reference-Python golden vectors."""
import os

import numpy as np
import pytest
import torch

from conftest import gen_panoptic_maps

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _run(seg, pan, cls, limit, S=19, C=9):
    from upsnet_amd import ops
    out = ops.unified_pan_result(torch.from_numpy(pan.astype(np.int64)).cuda(), torch.from_numpy(seg.astype(np.int64)).cuda(),
                                 torch.from_numpy(np.asarray(cls, np.int64)), S - C, S, limit)
    return out.cpu().numpy()


def test_unified_pan_golden_from_reference_python():
    g = np.load(os.path.join(G, "unified_pan.npz"))
    for tag in ("a", "b", "c"):
        np.testing.assert_array_equal(_run(g[tag + "_seg"], g[tag + "_pan"], g[tag + "_cls"], int(g[tag + "_limit"])), g[tag + "_out"])


@pytest.mark.parametrize("H,W,k,limit,seed", [(64, 96, 5, 100, 0), (128, 256, 40, 2000, 1), (33, 47, 3, 50, 2), (64, 64, 0, 10 ** 9, 3),
                                              (200, 300, 100, 4 * 64 * 64, 4)])
def test_unified_pan_vs_oracle(H, W, k, limit, seed):
    from oracle import ops as oops
    rng = np.random.default_rng(seed)
    seg, pan, cls = gen_panoptic_maps(rng, H, W, k)
    ref = oops.get_unified_pan_result(seg, pan, cls, 19, 9, limit)
    np.testing.assert_array_equal(_run(seg, pan, cls, limit), ref)


def test_unified_pan_full_size_and_module_api():
    """1024 x 2048 through the host mirror of BaseDataset, COCO class counts (133 seg / 81 det classes)."""
    from oracle import ops as oops
    from upsnet_amd.config.config import update_config_dict, COCO_R101_DCN, CITYSCAPES_R50
    from upsnet_amd.dataset.base_dataset import BaseDataset
    update_config_dict(COCO_R101_DCN)
    try:
        rng = np.random.default_rng(9)
        seg, pan, cls = gen_panoptic_maps(rng, 1024, 2048, 60, num_stuff=53, num_things=80)
        out = BaseDataset().get_unified_pan_result([torch.from_numpy(seg).cuda()[None]], [torch.from_numpy(pan).cuda()[None]], [cls])[0]
        ref = oops.get_unified_pan_result(seg, pan, cls, 133, 81)
        np.testing.assert_array_equal(out.cpu().numpy(), ref)
    finally:
        update_config_dict(CITYSCAPES_R50)


def test_unified_pan_on_model_output():
    """End of the chain: the network's own label maps -> 2-channel result == oracle on the same maps."""
    from oracle import ops as oops
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, config
    update_config_dict(CITYSCAPES_R50)
    from upsnet_amd.dataset.base_dataset import BaseDataset
    from upsnet_amd.synthetic import build_model, make_image
    model = build_model(cls_gain=0.3)
    with torch.no_grad():
        out = model(make_image(256, 512, seed=0, device='cuda'))
    res = BaseDataset().get_unified_pan_result([out['fcn_outputs']], [out['panoptic_outputs']], [out['panoptic_cls_inds']], stuff_area_limit=500)[0]
    ref = oops.get_unified_pan_result(out['fcn_outputs'][0].cpu().numpy(), out['panoptic_outputs'][0].cpu().numpy(),
                                      out['panoptic_cls_inds'].cpu().numpy(), config.dataset.num_seg_classes, config.dataset.num_classes, 500)
    np.testing.assert_array_equal(res.cpu().numpy(), ref)
