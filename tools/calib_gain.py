import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
update_config_dict(CITYSCAPES_R50)
from upsnet_amd.synthetic import build_model, make_image
for (H, W) in ((1024, 2048), (256, 512)):
    for seed in (0, 1):
        data = make_image(H, W, seed=seed, device='cuda')
        for gain in (0.0005, 0.001, 0.002, 0.005, 0.01, 0.02, 0.05, 0.1, 0.3):
            model = build_model(cls_gain=gain)
            with torch.no_grad():
                out = model(data)
                rc = model.rcnn(list(model.fpn(*model.resnet_backbone(data['data']))[:4]), torch.zeros(4, 5, device='cuda'))
            print(H, W, "seed", seed, "gain", gain, "n_det", out['cls_inds'].numel(), "n_inst", out['panoptic_cls_inds'].numel(),
                  "nlabels", torch.unique(out['panoptic_outputs']).numel(), "maxlogit %.3g" % rc['cls_score'].abs().max().item(), flush=True)
