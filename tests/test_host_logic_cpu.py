"""CPU: host-side logic of the product (no kernels): config, anchors, synthetic inputs, model structure,
state-dict key names (the checkpoint contract with the reference), BN folding, fc6 weight re-layout."""
import os

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(autouse=True)
def _cfg():
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)


def test_product_anchors_match_reference_golden():
    from upsnet_amd.rpn.generate_anchors import generate_anchors
    g = np.load(os.path.join(G, "anchors.npz"))
    for s in (4, 8, 16, 32, 64):
        assert np.array_equal(generate_anchors(s, (8 * s,), (0.5, 1, 2)), g["s%d" % s])


def test_product_bbox_helpers_match_golden():
    from upsnet_amd.bbox.bbox_transform import bbox_transform, clip_boxes
    g = np.load(os.path.join(G, "bbox_transform.npz"))
    np.testing.assert_allclose(bbox_transform(g["boxes"], g["deltas"], (10., 10., 5., 5.)), g["decoded"], rtol=2e-6, atol=2e-4)
    assert np.array_equal(clip_boxes(g["decoded"].copy(), (600, 900)), g["clipped"])


def test_config_yaml_roundtrip(tmp_path):
    from upsnet_amd.config.config import config, update_config
    p = tmp_path / "exp.yaml"
    p.write_text("symbol: resnet_101_upsnet\ndataset:\n  num_classes: 81\n  num_seg_classes: 133\nnetwork:\n  fcn_num_layers: 3\n  fpn_with_gap: true\ntest:\n  rpn_post_nms_top_n: 300\n")
    update_config(str(p))
    assert config.symbol == 'resnet_101_upsnet' and config.dataset.num_seg_classes == 133
    assert config.network.fpn_with_gap is True and config.test.rpn_post_nms_top_n == 300
    assert config.test.rpn_nms_thresh == 0.7 and config.network.bbox_reg_weights == (10., 10., 5., 5.)


def test_synthetic_image_contract():
    from upsnet_amd.synthetic import make_image
    d = make_image(800, 1333, seed=0)
    assert d['data'].shape == (1, 3, 800, 1344) and d['data'].dtype == torch.float32
    assert np.array_equal(d['im_info'], np.array([[800, 1333, 1.0]], np.float32))
    assert not d['data'][:, :, :, 1333:].any()
    px = d['data'][0, :, 0, 0] + torch.tensor([102.9801, 115.9465, 122.7717])
    assert torch.allclose(px, px.round(), atol=1e-4)
    assert torch.equal(make_image(64, 64, seed=3)['data'], make_image(64, 64, seed=3)['data'])


def test_model_structure_and_checkpoint_keys():
    from upsnet_amd.models.resnet_upsnet import resnet_50_upsnet
    m = resnet_50_upsnet()
    keys = set(m.state_dict().keys())
    # key names the reference checkpoints use (SURVEY.md section 5 "Checkpoint / resume")
    for k in ['resnet_backbone.conv1.conv1.weight', 'resnet_backbone.conv1.bn1.running_var',
              'resnet_backbone.res2.layers.0.downsample.0.weight', 'resnet_backbone.res2.layers.0.downsample.1.running_mean',
              'resnet_backbone.res3.layers.3.conv2.weight', 'resnet_backbone.res4.layers.5.bn3.bias', 'resnet_backbone.res5.layers.2.conv3.weight',
              'fpn.fpn_p5_1x1.weight', 'fpn.fpn_p2.bias', 'rpn.conv_proposal.0.weight', 'rpn.cls_score.weight', 'rpn.bbox_pred.bias',
              'rcnn.fc6.0.weight', 'rcnn.fc7.0.bias', 'rcnn.cls_score.weight', 'rcnn.bbox_pred.weight',
              'mask_branch.mask_conv1.0.weight', 'mask_branch.mask_conv4.0.bias', 'mask_branch.mask_deconv1.0.weight', 'mask_branch.mask_score.weight',
              'fcn_head.fcn_subnet.conv.0.0.conv_offset.weight', 'fcn_head.fcn_subnet.conv.0.0.conv.weight', 'fcn_head.fcn_subnet.conv.1.0.conv.bias',
              'fcn_head.score.weight']:
        assert k in keys, k
    assert m.state_dict()['rcnn.fc6.0.weight'].shape == (1024, 12544)
    assert m.state_dict()['fcn_head.fcn_subnet.conv.0.0.conv.weight'].shape == (128, 256, 3, 3)
    assert m.state_dict()['fcn_head.fcn_subnet.conv.1.0.conv.weight'].shape == (128, 128, 3, 3)
    assert m.state_dict()['rcnn.cls_score.weight'].shape == (9, 1024) and m.state_dict()['fcn_head.score.weight'].shape == (19, 512, 1, 1)
    n_blocks = [len(getattr(m.resnet_backbone, 'res%d' % i).layers) for i in (2, 3, 4, 5)]
    assert n_blocks == [3, 4, 6, 3]
    assert m.resnet_backbone.res3.layers[0].conv1.stride == (2, 2)   # caffe-style: stride on the first 1x1
    assert not m.fcn_head.fcn_subnet.conv[0][0].conv_offset.weight.any()  # zero-init offsets as the reference
    # tolerant loader with torchvision-style names
    sd = {'conv1.weight': torch.ones(64, 3, 7, 7), 'layer1.0.conv1.weight': torch.full((64, 64, 1, 1), 2.0), 'module.bogus': torch.zeros(1)}
    with pytest.warns(UserWarning):
        m.load_state_dict(sd)
    assert float(m.resnet_backbone.conv1.conv1.weight.mean()) == 1.0 and float(m.resnet_backbone.res2.layers[0].conv1.weight.mean()) == 2.0


def test_dcn_backbone_config():
    from upsnet_amd.config.config import update_config_dict, COCO_R101_DCN
    update_config_dict(COCO_R101_DCN)
    from upsnet_amd.models.resnet import ResNetBackbone, DCNBottleneck
    b = ResNetBackbone([3, 4, 23, 3])
    assert all(isinstance(l, DCNBottleneck) for l in b.res3.layers) and all(isinstance(l, DCNBottleneck) for l in b.res5.layers)
    assert not any(isinstance(l, DCNBottleneck) for l in b.res2.layers)
    assert 'res4.layers.22.conv2_offset.weight' in b.state_dict()
    # r13 routing rule: the plain 3x3 layers upstream of the deformable chain (res2) stay off the F(4x4) Winograd form
    assert all(l.feeds_deformable for l in b.res2.layers)
    assert not any(l.feeds_deformable for n in ('res3', 'res4', 'res5') for l in getattr(b, n).layers)
    from upsnet_amd.config.config import CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    b50 = ResNetBackbone([3, 4, 6, 3])
    assert not any(l.feeds_deformable for n in ('res2', 'res3', 'res4', 'res5') for l in getattr(b50, n).layers)


def test_fold_frozen_bn_is_exact_reparameterisation():
    from upsnet_amd.models.resnet import Bottleneck, fold_frozen_bn
    import torch.nn as nn
    torch.manual_seed(0)
    ds = nn.Sequential(nn.Conv2d(16, 32, 1, bias=False), nn.BatchNorm2d(32))
    blk = Bottleneck(16, 8, downsample=ds).eval()
    for bn in (blk.bn1, blk.bn2, blk.bn3, ds[1]):
        bn.weight.data.uniform_(0.5, 1.5); bn.bias.data.normal_(); bn.running_mean.normal_(); bn.running_var.uniform_(0.5, 2)
    x = torch.randn(2, 16, 9, 9)
    with torch.no_grad():
        y0 = blk(x.clone())
        y1 = fold_frozen_bn(blk)(x.clone())
    assert isinstance(blk.bn1, nn.Identity)
    torch.testing.assert_close(y0, y1, rtol=1e-4, atol=1e-5)


def test_fc6_weight_relayout_matches_nchw_flatten():
    from upsnet_amd.models.rcnn import RCNN
    torch.manual_seed(1)
    r = RCNN(9, 9, dim_in=8, dim_hidden=16)
    pooled = torch.randn(5, 8, 7, 7)
    ref = torch.nn.functional.linear(pooled.reshape(5, -1), r.fc6[0].weight, r.fc6[0].bias)
    nhwc = pooled.contiguous(memory_format=torch.channels_last)
    got = torch.nn.functional.linear(nhwc.permute(0, 2, 3, 1).reshape(5, -1), r._fc6_weight_nhwc(), r.fc6[0].bias)
    torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)


def test_blob_geometry_host_mirror_equals_oracle():
    """dataset/blob.py restates prep_im_for_blob's scale rule + im_list_to_blob's padding exactly like the oracle (base_dataset.py:155-170, 909-913)."""
    import oracle
    from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
    update_config_dict(CITYSCAPES_R50)
    from upsnet_amd.dataset.blob import blob_geometry
    for (H, W, target, max_size) in [(1024, 2048, 1024, 2048), (480, 640, 800, 1333), (427, 640, 800, 1333), (1333, 500, 800, 1333),
                                     (375, 1242, 800, 1333), (50, 75, 50, 100), (37, 53, 64, 96)]:
        assert blob_geometry(H, W, target, max_size) == oracle.blob_geometry(H, W, target, max_size), (H, W)
    # the COCO setting: short side 800 unless the long side would exceed 1333
    s, (hr, wr), (hp, wp) = blob_geometry(480, 640, 800, 1333)
    assert abs(s - 800 / 480) < 1e-12 and (hr, wr) == (800, 1067) and (hp, wp) == (800, 1088)
    s, (hr, wr), _ = blob_geometry(375, 1242, 800, 1333)
    assert abs(s - 1333 / 1242) < 1e-12 and wr == 1333


def test_conv_instance_selection_rules():
    """hipconv picks the kernel instance by shape only (never by data): Winograd for 3x3/s1 layers with >= 128 workgroups of
    2x2-tile work (64-tile workgroups above 768 of them, else 32-tile ones; its own K split for single maps below 256
    workgroups), split-K direct for <= 256 direct workgroups with a long K walk, direct otherwise."""
    import torch
    import torch.nn as nn
    from upsnet_amd.models import hipconv
    c3 = nn.Conv2d(256, 256, 3, padding=1)
    c5 = nn.Conv2d(512, 512, 3, padding=1)
    x = lambda n, h, w, c=256: torch.empty(n, c, h, w, device='meta')
    assert hipconv._use_winograd(c3, [x(1, 256, 512)]) and hipconv._use_winograd(c3, [x(1, 128, 256)])      # FPN P2, P3
    assert hipconv._wino_tm(c3, [x(1, 256, 512)]) == 64 and hipconv._wino_tm(c3, [x(1, 128, 256)]) == 32     # 2048 / 512 workgroups of 64 tiles
    assert hipconv._wino_ksplit(c3, [x(1, 256, 512)]) == 1 and hipconv._wino_ksplit(c3, [x(1, 128, 256)]) == 1
    assert hipconv._use_winograd(c3, [x(1, 64, 128)]) and hipconv._wino_ksplit(c3, [x(1, 64, 128)]) == 1    # P4 / res4: 256 x 32-tile wgs
    assert hipconv._use_winograd(c5, [x(1, 32, 64, 512)]) and hipconv._wino_ksplit(c5, [x(1, 32, 64, 512)]) == 4   # res5: 128 x 4
    assert hipconv._use_winograd(c3, [x(1, 32, 64)]) and hipconv._wino_ksplit(c3, [x(1, 32, 64)]) == 4      # P5: 64 x 4 (4 slabs each)
    assert not hipconv._use_winograd(c3, [x(1, 16, 32)])                                                      # P6: too few tiles
    assert hipconv._use_winograd(c3, [x(1, 256 >> l, 512 >> l) for l in range(5)])                            # RPN over 5 levels
    assert hipconv._use_winograd(nn.Conv2d(256, 18, 3, padding=1), [x(1, 256 >> l, 512 >> l) for l in range(4)])   # DCN offsets
    assert not hipconv._use_winograd(nn.Conv2d(256, 256, 3, stride=2, padding=1), [x(1, 256, 512)])           # strided
    assert not hipconv._use_winograd(nn.Conv2d(256, 256, 3, padding=2, dilation=2), [x(1, 256, 512)])         # dilated
    assert not hipconv._use_winograd(nn.Conv2d(256, 256, 1), [x(1, 256, 512)])
    assert hipconv._use_winograd(c3, [x(3, 14, 14)], always=True) and not hipconv._use_winograd(c3, [x(3, 14, 14)])   # mask head: pinned
    assert hipconv._ksplit(nn.Conv2d(512, 512, 3, padding=1), x(1, 32, 64, 512), 512) == 3                    # res5 conv2: 256 workgroups
    assert hipconv._ksplit(c3, x(1, 32, 64), 256) == 4                                                        # FPN P5: 128 workgroups
    assert hipconv._ksplit(c3, x(1, 64, 128), 256) == 1 and hipconv._ksplit(nn.Conv2d(256, 1024, 1), x(1, 64, 128), 1024) == 1


def test_mask_head_tail_split_rule(monkeypatch):
    """hipconv, r10: the mask head's batched Winograd launch is split into main + half-size tail only where the tail fits one half-size
    workgroup per CU. Shape-only decisions."""
    import torch
    import torch.nn as nn
    from upsnet_amd.models import hipconv
    monkeypatch.setattr(hipconv, '_cus', lambda device: 256)
    monkeypatch.setattr(hipconv, 'WINO_TAIL_SPLIT', True)
    c3 = nn.Conv2d(256, 256, 3, padding=1)
    x = lambda n, h=14, w=14, c=256: torch.empty(n, c, h, w, device='meta')
    assert hipconv._wino_tail_split(c3, x(100)) == 80          # 616 workgroups for 512 slots: 80 ROIs + 20 on half-size workgroups (248 <= 256)
    assert hipconv._wino_tail_split(c3, x(83)) == 0            # exactly one round
    assert hipconv._wino_tail_split(c3, x(64)) == 0 and hipconv._wino_tail_split(c3, x(1)) == 0
    assert hipconv._wino_tail_split(c3, x(143)) == 0           # the tail (60 ROIs) would be 736 half-size workgroups
    assert hipconv._wino_tail_split(c3, x(200)) == 0
    assert hipconv._wino_tail_split(nn.Conv2d(256, 96, 3, padding=1), x(100)) == 0     # Cout % 64
    monkeypatch.setattr(hipconv, 'WINO_TAIL_SPLIT', False)
    assert hipconv._wino_tail_split(c3, x(100)) == 0


def test_winograd_f4x4_routing_rule(monkeypatch):
    """hipconv, r12: the F(4x4,3x3) kernel only where a launch gives every CU a 32-tile x 64-channel workgroup and its last round is at least
    UPSNET_WINO36_MIN_FILL full; never bf16 inputs, Cout % 64 != 0, Cin % 32 != 0. Shape-only decisions."""
    import torch
    import torch.nn as nn
    from upsnet_amd.models import hipconv
    monkeypatch.setattr(hipconv, '_cus', lambda device: 256)
    monkeypatch.setattr(hipconv, 'WINO36', True)
    monkeypatch.setattr(hipconv, 'WINOGRAD', True)
    monkeypatch.setattr(hipconv, 'WINO36_MIN_FILL', 0.65)
    c3 = nn.Conv2d(256, 256, 3, padding=1)
    x = lambda n, h, w, c=256, dt=torch.float32: torch.empty(n, c, h, w, device='meta', dtype=dt)
    use = hipconv._use_winograd36
    assert use(c3, [x(1, 256, 512)]) and use(c3, [x(1, 128, 256)])                       # FPN P2 (1024 workgroups), P3 (256)
    assert not use(c3, [x(1, 64, 128)]) and not use(c3, [x(1, 32, 64)])                   # P4 / res4 (64), P5
    assert use(c3, [x(1, 256 >> l, 512 >> l) for l in range(5)])                           # RPN over 5 levels: 1364 workgroups, 0.89
    assert use(nn.Conv2d(64, 64, 3, padding=1), [x(1, 256, 512, 64)])                      # res2 conv2: 256
    assert not use(nn.Conv2d(128, 128, 3, padding=1), [x(1, 128, 256, 128)])               # res3 conv2: 128
    assert use(c3, [x(1, 200, 336)])                                                       # UPSNet-101-DCN 800x1333 FPN P2: 528 of 768 = 0.69
    assert use(c3, [x(1, 200, 336), x(1, 100, 168), x(1, 50, 84), x(1, 25, 42), x(1, 13, 21)])
    assert not use(c3, [x(1, 132, 256)])                                                   # 264 workgroups: a second round 3 % full
    assert use(c3, [x(1, 100, 168)])                                                       # r13: UPSNet-101-DCN 800x1333 FPN P3, 132 workgroups = half a round, 16 slabs: 94.9 vs 120.6 us
    assert hipconv._wino36_ksplit(c3, x(1, 64, 128)) == 4 and hipconv._wino36_ksplit(nn.Conv2d(128, 128, 3, padding=1), x(1, 128, 256, 128)) == 2   # (opt-in split-K form)
    assert hipconv._wino36_ksplit(c3, x(1, 100, 168)) == 1 and hipconv._wino36_ksplit(c3, x(1, 32, 64)) == 1 and hipconv._wino36_ksplit(c3, x(1, 256, 512)) == 1
    assert not use(nn.Conv2d(256, 18, 3, padding=1), [x(1, 256, 512)])                     # offset predictors: Cout % 64
    assert not use(nn.Conv2d(48, 64, 3, padding=1), [x(1, 256, 512, 48)])                  # Cin % 32
    assert not use(c3, [x(1, 256, 512, dt=torch.bfloat16)])
    assert not use(nn.Conv2d(256, 256, 3, stride=2, padding=1), [x(1, 256, 512)]) and not use(nn.Conv2d(256, 256, 1), [x(1, 256, 512)])
    # r13 (opt-in, UPSNET_WINO36_PREFIX): a multi-map launch whose last round of workgroups is less than half full keeps only the leading maps that make whole rounds
    monkeypatch.setattr(hipconv, 'WINO36_PREFIX', True)
    pre = hipconv._wino36_round_prefix
    assert pre(c3, [x(1, 256 >> l, 512 >> l) for l in range(5)]) == 2       # RPN at 1024x2048: 1024 + 256 = 5 rounds | 64 + 16 + 4 would be a third of a sixth
    assert pre(c3, [x(1, 200, 336), x(1, 100, 168), x(1, 50, 84), x(1, 25, 42), x(1, 13, 21)]) == 5      # 712 workgroups: last round 0.78 full
    assert pre(c3, [x(1, 256, 512)]) == 1 and pre(c3, [x(1, 64 >> l, 128 >> l) for l in range(5)]) == 5  # (no prefix makes a whole round: nothing to cut)
    monkeypatch.setattr(hipconv, 'WINO36', False)
    assert not use(c3, [x(1, 256, 512)])


def test_small_tile_1x1_routing_rule(monkeypatch):
    """hipconv, r13: the 16x16x4-fragment 1x1 kernel only where the 64-pixel tiles leave the last round of workgroups < 0.75 full AND the
    layer is a long K walk into few channels (Cin >= 2 Cout, Cin >= 256). Shape-only; the headline's power-of-two maps never take it."""
    import torch
    import torch.nn as nn
    from upsnet_amd.models import hipconv
    monkeypatch.setattr(hipconv, 'KSW', True)
    monkeypatch.setattr(hipconv, 'KSW_MAX_FILL', 0.75)
    monkeypatch.setattr(hipconv, 'PRECISION', 'fp32')
    x = lambda c, h, w, n=1: torch.empty(n, c, h, w, device='meta')
    use = lambda cin, cout, h, w, st=1: hipconv._use_ksw(nn.Conv2d(cin, cout, 1, stride=st), x(cin, h, w), cus=256)
    assert use(1024, 256, 50, 84)            # UPSNet-101-DCN res4 conv1 at 800x1333: 264 workgroups of 512 slots
    assert use(2048, 512, 25, 42) and use(2048, 256, 25, 42)      # res5 conv1 (136 of 256), the P5 lateral (68 of 256)
    assert use(512, 256, 100, 168, 2)        # a stage's first conv1 (stride 2) when it runs alone
    assert use(512, 128, 100, 168)           # res3 conv1: 526 of 768
    assert not use(256, 1024, 50, 84) and not use(512, 2048, 25, 42) and not use(128, 512, 100, 168)   # conv3 layers: Cin < 2 Cout
    assert not use(256, 64, 200, 336)        # res2 conv1: 1050 of 1280 = 0.82 full
    for cin, cout, h, w in [(1024, 256, 64, 128), (2048, 512, 32, 64), (512, 128, 128, 256), (256, 64, 256, 512)]:
        assert not use(cin, cout, h, w), (cin, cout, h, w)       # the headline workload (1024x2048): its bottleneck maps tile evenly
    assert use(2048, 256, 32, 64)            # ... its P5 lateral is 128 workgroups for 256 CUs: 29.2 -> 21.0 us (was the general kernel, split-K x4)
    assert not hipconv._use_ksw(nn.Conv2d(1024, 256, 1), torch.empty(1, 1024, 50, 84, device='meta', dtype=torch.bfloat16), cus=256)
    monkeypatch.setattr(hipconv, 'KSW', False)
    assert not use(1024, 256, 50, 84)


def test_knobs_report_set_variables_and_reject_unknown_names():
    """bench hygiene: every UPSNET_* variable that is set goes into the bench line; a name no source file reads is an error."""
    from upsnet_amd import knobs
    names = knobs.known()
    assert {'UPSNET_GRAPH', 'UPSNET_OVERLAP', 'UPSNET_WINOGRAD', 'UPSNET_CONV_PRECISION', 'UPSNET_DECONV_FRAG', 'UPSNET_ROI_KERNEL',
            'UPSNET_NMS_SCAN16', 'UPSNET_SHARE_GPU'} <= names
    env = {'PATH': '/bin', 'UPSNET_GRAPH': '0', 'UPSNET_DECONV_FRAG': '0'}
    assert knobs.active(env) == {'UPSNET_DECONV_FRAG': '0', 'UPSNET_GRAPH': '0'} and knobs.check(env) == knobs.active(env)
    with pytest.raises(ValueError) as e:
        knobs.check({'UPSNET_GRAHP': '0'})
    assert 'UPSNET_GRAHP' in str(e.value)
    assert knobs.check({'HOME': '/root'}) == {}
    # the explicit registry is what check() trusts (installed layouts have no .hip sources to scan); it must equal the source scan
    assert knobs.known() == knobs.scan_sources(), (knobs.known() ^ knobs.scan_sources())
    assert all(len(v) == 3 for v in knobs.REGISTRY.values())
