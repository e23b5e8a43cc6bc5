"""Instance masks -> COCO RLE: host mirror of im_post (upsnet/upsnet_end2end_test.py:95-152).

The reference resizes / thresholds / pastes every detection's mask over the whole image on the host and calls
pycocotools.mask.encode. Here the device kernel (csrc/postprocess.hip: im_post_rle_kernel) evaluates only the box region and
returns the run-length encoding as the list of column-major pixel indices where the mask value changes; this module turns
that into pycocotools' `counts` (differences) and its compressed string (rleToString of pycocotools' maskApi.c -- a third-party
format, restated from the published algorithm).
"""

import numpy as np
import torch

from .._lib import check, lib, ptr, stream


def counts_from_transitions(positions, num_pixels):
    """Sorted change positions -> pycocotools counts (alternating zeros / ones, starting with zeros)."""
    p = np.concatenate([[0], np.asarray(positions, np.int64), [int(num_pixels)]])
    return np.diff(p).tolist()


def rle_to_string(counts):
    out = []
    for i, c in enumerate(counts):
        x = int(c)
        if i > 2:
            x -= int(counts[i - 2])
        more = True
        while more:
            ch = x & 0x1f
            x >>= 5
            more = (x != -1) if (ch & 0x10) else (x != 0)
            if more:
                ch |= 0x20
            out.append(chr(ch + 48))
    return ''.join(out)


def mask_transitions(pred_boxes, pred_masks, cls_inds, im_h, im_w, cap=4096):
    """Device part: pred_boxes [n,4] (image coordinates), pred_masks [n,C,M,M] (or [n,1,M,M]), cls_inds [n] ->
    list of n int64 arrays of change positions. Grows `cap` and retries when a mask has more transitions."""
    n = pred_boxes.shape[0]
    if n == 0:
        return []
    boxes = pred_boxes.float().contiguous()
    masks = pred_masks.float().contiguous()
    cls = cls_inds.to(torch.int64).contiguous()
    if not (boxes.is_cuda and masks.is_cuda and cls.is_cuda):
        raise RuntimeError("mask_transitions: CUDA tensors required")
    C, M = masks.shape[1], masks.shape[-1]
    while True:
        trans = torch.empty((n, cap), dtype=torch.int32, device=boxes.device)
        cnt = torch.empty((n,), dtype=torch.int32, device=boxes.device)
        check(lib().upsnet_im_post_rle(stream(), ptr(boxes), ptr(masks), ptr(cls), n, C, M, int(im_h), int(im_w), int(cap), ptr(trans), ptr(cnt)),
              "im_post_rle")
        c = cnt.cpu().numpy()
        if int(c.max()) <= cap:
            break
        cap = int(c.max())
    t = trans.cpu().numpy().view(np.uint32)
    return [t[d, :c[d]].astype(np.int64) for d in range(n)]


def im_post(boxes_all, masks_all, scores, pred_boxes, pred_masks, cls_inds, num_classes, im_info):
    """Same contract as the reference's im_post: appends per-class [boxes|score] arrays to boxes_all[cls] and lists of
    {'size': [h, w], 'counts': str} to masks_all[cls]. pred_* are device tensors of ONE image; im_info = (height, width)."""
    im_h, im_w = int(im_info[0]), int(im_info[1])
    trans = mask_transitions(pred_boxes, pred_masks, cls_inds, im_h, im_w)
    cls_np = cls_inds.cpu().numpy()
    boxes_np = pred_boxes.float().cpu().numpy()
    scores_np = np.asarray(scores.cpu().numpy() if isinstance(scores, torch.Tensor) else scores).reshape(-1, 1)
    for idx in range(1, num_classes):
        sel = np.nonzero(cls_np == idx)[0]
        segms = [{'size': [im_h, im_w], 'counts': rle_to_string(counts_from_transitions(trans[d], im_h * im_w))} for d in sel]
        boxes_all[idx].append(np.hstack([boxes_np[sel], scores_np[sel]]))
        masks_all[idx].append(segms)
