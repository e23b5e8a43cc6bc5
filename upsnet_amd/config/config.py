"""Global configuration of the inference path.

Mirrors upsnet/config/config.py:20-198 of the reference (same field names and defaults, same
``update_config(yaml)`` entry point) without the easydict dependency. Only the fields that
parameterise the inference hot path are kept (SURVEY.md section 5 "Config / flags").
"""
import numpy as np
import yaml


class AttrDict(dict):
    """Minimal easydict stand-in: attribute access, nested dicts converted on assignment."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = AttrDict(v) if isinstance(v, dict) and not isinstance(v, AttrDict) else v


config = AttrDict()
config.symbol = 'resnet_50_upsnet'
config.gpus = '0'

config.network = AttrDict()
config.network.backbone_fix_bn = True
config.network.backbone_with_dilation = False
config.network.backbone_with_dpyramid = False
config.network.backbone_with_dconv = 100
config.network.backbone_freeze_at = 2
config.network.use_caffe_model = True
config.network.use_syncbn = False
config.network.has_rcnn = True
config.network.has_mask_head = True
config.network.has_fcn_head = True
config.network.has_panoptic_head = True
config.network.pixel_means = np.array((102.9801, 115.9465, 122.7717,))
config.network.cls_agnostic_bbox_reg = False
config.network.rcnn_feat_stride = 32
config.network.bbox_reg_weights = (10., 10., 5., 5.,)
config.network.rpn_feat_stride = (4, 8, 16, 32, 64,)
config.network.anchor_ratios = (0.5, 1, 2)
config.network.anchor_scales = (8,)
config.network.num_anchors = 3
config.network.rpn_with_norm = 'none'
config.network.has_fpn = True
config.network.fpn_feature_dim = 256
config.network.fpn_with_gap = False
config.network.fpn_upsample_method = 'nearest'
config.network.fpn_with_norm = 'none'
config.network.rcnn_with_norm = 'none'
config.network.mask_size = 28
config.network.binary_thresh = 0.5
config.network.has_mask_rcnn = True
config.network.fcn_with_norm = 'none'
config.network.fcn_num_layers = 3
config.network.fcn_head = 'FCNHead'

config.dataset = AttrDict()
config.dataset.num_classes = 9        # Cityscapes defaults (upsnet_resnet50_cityscapes_16gpu.yaml)
config.dataset.num_seg_classes = 19

config.train = AttrDict()
config.train.use_horovod = False
config.train.rpn_individual_proposals = True
config.train.panoptic_box_keep_fraction = 0.7
config.train.fcn_with_roi_loss = False

config.test = AttrDict()
config.test.rpn_nms_thresh = 0.7
config.test.rpn_pre_nms_top_n = 1000
config.test.rpn_post_nms_top_n = 1000
config.test.rpn_min_size = 0
config.test.nms_thresh = 0.5
config.test.max_det = 100
config.test.score_thresh = 0.05
config.test.panoptic_score_thresh = 0.6
config.test.panoptic_stuff_area_limit = 4096
config.test.batch_size = 1
config.test.scales = [1024]
config.test.max_size = 2048


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict):
            if k not in dst or not isinstance(dst[k], dict):
                dst[k] = AttrDict()
            _merge(dst[k], v)
        else:
            if k in ('bbox_reg_weights', 'anchor_ratios', 'anchor_scales', 'rpn_feat_stride') and isinstance(v, list):
                v = tuple(v)
            dst[k] = v


def update_config(config_file):
    """config.py:177-198: merge a reference experiment yaml into the global config."""
    with open(config_file) as f:
        exp = yaml.safe_load(f)
    _merge(config, exp)
    return config


def update_config_dict(d):
    _merge(config, d)
    return config


# experiment presets equivalent to the reference yamls that matter for the benchmark configs
CITYSCAPES_R50 = dict(symbol='resnet_50_upsnet', dataset=dict(num_classes=9, num_seg_classes=19),
                      network=dict(fcn_num_layers=2, fpn_with_gap=False, backbone_with_dconv=100),
                      test=dict(rpn_post_nms_top_n=1000, rpn_pre_nms_top_n=1000, scales=[1024], max_size=2048),
                      train=dict(panoptic_box_keep_fraction=0.7))
COCO_R101_DCN = dict(symbol='resnet_101_upsnet', dataset=dict(num_classes=81, num_seg_classes=133),
                     network=dict(fcn_num_layers=3, fpn_with_gap=True, backbone_with_dconv=3),
                     test=dict(rpn_post_nms_top_n=300, rpn_pre_nms_top_n=1000, scales=[800], max_size=1333),
                     train=dict(panoptic_box_keep_fraction=0.7))
