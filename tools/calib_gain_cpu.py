"""Host-only sweep of synthetic.DEFAULT_CLS_GAIN: trunk + proposals + box head once on the CPU (oracle.forward pieces), then the two
detection selections + the panoptic keep count as a function of the classifier gain. Usage: python tools/calib_gain_cpu.py [c1|c2] [H W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn.functional as F
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, COCO_R101_DCN, config
which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
update_config_dict(CITYSCAPES_R50 if which == 'c1' else COCO_R101_DCN)
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else ((1024, 2048) if which == 'c1' else (800, 1333))
from upsnet_amd.synthetic import build_model, make_image
from oracle import ops as oops
from oracle.forward import _backbone_cpu, _fpn_pool_cpu, _np
torch.set_num_threads(os.cpu_count())
t0 = time.time()
m = build_model(cls_gain=None, device='cpu', channels_last=False)
print("build %.1fs" % (time.time() - t0), flush=True)
cfg = config
C = cfg.dataset.num_classes
for seed in (0, 1):
    data = make_image(H, W, seed=seed)
    with torch.no_grad():
        t0 = time.time()
        res = _backbone_cpu(m.resnet_backbone, data['data'])
        pyr = m.fpn(*res)
        print("trunk %.1fs" % (time.time() - t0), "res rms", [round(float(r.pow(2).mean().sqrt()), 2) for r in res], "max", [round(float(r.abs().max()), 1) for r in res],
              "pyr rms", [round(float(r.pow(2).mean().sqrt()), 2) for r in pyr], flush=True)
        probs, boxes = [], []
        for f in pyr:
            _, b, p = m.rpn(f)
            probs.append(_np(p)); boxes.append(_np(b))
        for i in range(m.fcn_head.fcn_subnet.num_layers if seed == 0 else 0):
            off = m.fcn_head.fcn_subnet.conv[i][0].conv_offset(pyr[0] if i == 0 else pyr[0][:, :m.fcn_head.fcn_subnet.conv[i][0].conv_offset.in_channels])
            print("  fcn offset layer", i, "std %.2f px max %.1f" % (float(off.std()), float(off.abs().max())))
        im_info = np.asarray(data['im_info'], np.float32)
        rois, _ = oops.pyramid_proposal(probs, boxes, im_info[0], cfg.network.rpn_feat_stride, cfg.network.anchor_scales, cfg.network.anchor_ratios,
                                        cfg.test.rpn_pre_nms_top_n, cfg.test.rpn_post_nms_top_n, cfg.test.rpn_nms_thresh, cfg.test.rpn_min_size)
        feats = list(pyr[:4])
        pool = _fpn_pool_cpu(feats, rois, 7)
        fc7 = m.rcnn.fc7(F.relu(m.rcnn.fc6[0](pool.reshape(pool.shape[0], -1))))
        score = m.rcnn.cls_score(fc7)
        bbox_pred = _np(m.rcnn.bbox_pred(fc7))
        print("n_rois", rois.shape[0], "fc7 rms %.2f" % float(fc7.pow(2).mean().sqrt()), "cls logit std %.4f" % float(score.std()), flush=True)
        for gain in (4, 5, 6, 7, 8, 9, 10, 12, 20, 25, 30, 35):
            cp = _np(F.softmax(score * gain, dim=1))
            ds, db, dc = oops.mask_roi(rois, bbox_pred, cp, im_info, C, cfg.test.nms_thresh, cfg.test.score_thresh, cfg.test.max_det, False, cfg.network.bbox_reg_weights)
            ps, pb, pc = oops.mask_roi(rois, bbox_pred, cp, im_info, C, 0.5, cfg.test.panoptic_score_thresh, cfg.test.max_det, True, cfg.network.bbox_reg_weights)
            print("  seed", seed, "gain", gain, "n_det", db.shape[0], "n_pan", pb.shape[0], "maxp %.3f" % cp[:, 1:].max(), flush=True)
