"""Post-processing of the network outputs -- host mirror of BaseDataset.get_unified_pan_result
(upsnet/dataset/base_dataset.py:332-371), running on the device (csrc/postprocess.hip)."""
import torch

from .. import ops
from ..config.config import config


class BaseDataset(object):
    """Only the result-shaping methods of the reference class that sit directly after the inference path."""

    def get_unified_pan_result(self, segs, pans, cls_inds, stuff_area_limit=4 * 64 * 64):
        """segs / pans: per-image int64 label maps ([H,W] or [1,H,W], device tensors; numpy arrays are uploaded);
        cls_inds: per-image 1-based thing classes of the panoptic instances. Returns a list of uint8 [H,W,3] device tensors
        (channel 0 category, channel 1 instance id), like the reference's list of numpy arrays."""
        id_last_stuff = config.dataset.num_seg_classes - config.dataset.num_classes
        out = []
        for seg, pan, cls_ind in zip(segs, pans, cls_inds):
            dev = pan.device if isinstance(pan, torch.Tensor) and pan.is_cuda else torch.device('cuda', torch.cuda.current_device())
            seg, pan, cls_ind = (torch.as_tensor(t).to(dev) for t in (seg, pan, cls_ind))
            out.append(ops.unified_pan_result(pan, seg, cls_ind, id_last_stuff, config.dataset.num_seg_classes, stuff_area_limit))
        return out
