"""GPU-side per-stage timing probe (development aid)."""
import sys, os, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
update_config_dict(CITYSCAPES_R50)
from upsnet_amd.synthetic import build_model, make_image

H, W = 1024, 2048
data = make_image(H, W, seed=0, device='cuda')
print("torch", torch.__version__, torch.cuda.get_device_name(0), "cpus", os.cpu_count(), "ref exists", os.path.isdir('/root/reference'), flush=True)
# stage timing with events
model = build_model()
with torch.no_grad():
    for _ in range(3):
        model(data)
torch.cuda.synchronize()
import upsnet_amd.models.resnet_upsnet as RU
def timed(fn, *a, n=5, **k):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn(*a, **k)
    torch.cuda.synchronize()
    return r, (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    x = data['data'].contiguous(memory_format=torch.channels_last)
    res, t = timed(model.resnet_backbone, x); print("backbone %.2f ms" % t)
    pyr, t = timed(model.fpn, *res); print("fpn %.2f ms" % t)
    def rpn_all():
        return [model.rpn(f) for f in pyr]
    r, t = timed(rpn_all); print("rpn %.2f ms" % t)
    probs = [a[2] for a in r]; boxes = [a[1] for a in r]
    pp, t = timed(model.pyramid_proposal.forward_padded, probs, boxes, data['im_info']); print("proposals %.3f ms" % t, "n_rois", int(pp[2].item()))
    feats = list(pyr[:4])
    fo, t = timed(model.fcn_head, *feats); print("fcn_head %.2f ms" % t)
    def offs():
        return [model.fcn_head.fcn_subnet.conv[0][0].conv_offset(f) for f in feats]
    o, t = timed(offs); print("  offset convs L0 %.2f ms" % t)
    from upsnet_amd import ops
    sub = model.fcn_head.fcn_subnet; dc = sub.conv[0][0].conv
    y, t = timed(ops.deform_conv_fused, feats, o, sub._wpack(0, dc), dc.bias, dc.in_channels, dc.out_channels, dc.kernel_size, dc.stride, dc.padding, dc.dilation, relu=True)
    print("  fused DCN L0 (4 levels) %.3f ms" % t)
    rc, t = timed(model.rcnn, feats, pp[0], pp[2]); print("rcnn %.2f ms" % t)
    pool, t = timed(model.rcnn.roi_pooling, feats, pp[0], pp[2]); print("  fpn_roi_align 7x7 x1000 %.3f ms" % t)
    cp = torch.softmax(rc['cls_score'], 1)
    mr, t = timed(model.mask_roi.forward_padded, pp[0], rc['bbox_pred'], cp, data['im_info'], pp[2]); print("mask_roi %.3f ms" % t)
    mr2, t = timed(model.mask_roi_panoptic.forward_padded, pp[0], rc['bbox_pred'], cp, data['im_info'], pp[2]); print("mask_roi_panoptic %.3f ms" % t)
    n1, n2 = int(mr[4].item()), int(mr2[4].item())
    both = torch.cat([mr[0][:n1], mr2[0][:n2]])
    ms_, t = timed(model.mask_branch, feats, both); print("mask_branch (%d rois) %.2f ms" % (both.shape[0], t))
    pl = ms_[n1:].gather(1, mr2[2][:n2].view(-1, 1, 1, 1).expand(-1, -1, 28, 28))
    fo_ = fo['fcn_output']
    sel, t = timed(model.mask_removal.select, mr2[0][:n2, 1:], mr2[1][:n2], pl, mr2[2][:n2], fo_.shape[2:]); print("mask_removal %.3f ms" % t)
    pf, t = timed(ops.panoptic_fuse, fo_, 11, mr2[0][:n2], pl, mr2[2][:n2], sel[0], sel[1], sel[2], model.seg_term.class_map, True); print("panoptic_fuse %.3f ms" % t)
