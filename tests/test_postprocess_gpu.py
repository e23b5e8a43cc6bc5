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
    model = build_model()
    with torch.no_grad():
        out = model(make_image(256, 512, seed=0, device='cuda'))
    res = BaseDataset().get_unified_pan_result([out['fcn_outputs']], [out['panoptic_outputs']], [out['panoptic_cls_inds']], stuff_area_limit=500)[0]
    ref = oops.get_unified_pan_result(out['fcn_outputs'][0].cpu().numpy(), out['panoptic_outputs'][0].cpu().numpy(),
                                      out['panoptic_cls_inds'].cpu().numpy(), config.dataset.num_seg_classes, config.dataset.num_classes, 500)
    np.testing.assert_array_equal(res.cpu().numpy(), ref)


def _dets(rng, n, H, W, C=9, M=28, num_classes=9):
    """Detections that exercise every paste case: inside, clipped at each border, full height, full width, tiny, outside."""
    from conftest import gen_rois
    boxes = gen_rois(rng, n, H, W, smin=4, smax=max(H, W))[:, 1:].astype(np.float32)
    boxes[0] = [-20.5, -15.2, W + 30.0, H + 12.0]          # covers the whole image
    boxes[1] = [10.3, -5.0, 40.7, H + 5.0]                 # full height, touches row H-1
    boxes[2] = [-8.0, 12.1, W + 4.0, 30.9]                 # full width
    boxes[3] = [W - 6.0, H - 9.0, W + 50.0, H + 70.0]      # bottom-right corner, mostly outside
    boxes[4] = [5.2, 7.7, 5.9, 8.1]                        # sub-pixel box
    boxes[5] = [W + 10.0, H + 10.0, W + 60.0, H + 40.0]    # completely outside
    masks = rng.random((n, C, M, M)).astype(np.float32)
    masks[0] = 0.9                                          # all ones over the whole image
    masks[1, :, :, :] = (np.arange(M)[None, :, None] > M // 2) * 0.95   # lower half set: runs end at row H-1
    cls = rng.integers(1, num_classes, size=n).astype(np.int64)
    return boxes, masks, cls


@pytest.mark.parametrize("H,W,n,seed", [(64, 96, 12, 0), (120, 75, 20, 1), (1024, 2048, 16, 2)])
def test_im_post_rle_vs_oracle(H, W, n, seed):
    """Device RLE (transition lists) == pycocotools-style run lengths of the oracle's pasted full-image masks, bit for bit."""
    from oracle import ops as oops
    from upsnet_amd.dataset.rle import counts_from_transitions, mask_transitions, rle_to_string
    rng = np.random.default_rng(seed)
    boxes, masks, cls = _dets(rng, n, H, W)
    ref = oops.im_post(boxes, masks, cls, H, W)
    trans = mask_transitions(torch.from_numpy(boxes).cuda(), torch.from_numpy(masks).cuda(), torch.from_numpy(cls).cuda(), H, W, cap=64)
    assert len(trans) == n
    for d in range(n):
        want = oops.rle_counts(ref[d])
        got = counts_from_transitions(trans[d], H * W)
        assert got == want, (d, boxes[d])
        assert rle_to_string(got) == oops.rle_to_string(want)
    assert ref[0].all() and not ref[5].any()


def test_im_post_module_api_class_agnostic_masks():
    """Host mirror of im_post (per-class lists of boxes+scores and RLE dicts), class-agnostic mask channel."""
    from oracle import ops as oops
    from upsnet_amd.dataset.rle import im_post
    rng = np.random.default_rng(7)
    H, W, n, C = 80, 100, 10, 9
    boxes, masks, cls = _dets(rng, n, H, W, C=1)
    scores = rng.random(n).astype(np.float32)
    boxes_all, masks_all = [[] for _ in range(C)], [[] for _ in range(C)]
    im_post(boxes_all, masks_all, torch.from_numpy(scores).cuda(), torch.from_numpy(boxes).cuda(), torch.from_numpy(masks).cuda(),
            torch.from_numpy(cls).cuda(), C, (H, W))
    ref = oops.im_post(boxes, masks, cls, H, W)
    seen = 0
    for idx in range(1, C):
        sel = np.nonzero(cls == idx)[0]
        assert len(boxes_all[idx]) == 1 and boxes_all[idx][0].shape == (len(sel), 5)
        np.testing.assert_array_equal(boxes_all[idx][0][:, :4], boxes[sel])
        for d, seg in zip(sel, masks_all[idx][0]):
            assert seg['size'] == [H, W] and seg['counts'] == oops.rle_to_string(oops.rle_counts(ref[d]))
            seen += 1
    assert seen == n
