"""Host-side box helpers with the reference's names (upsnet/bbox/bbox_transform.py:45-60,290-330).

The hot path decodes boxes on the device (csrc/common.h: ups_decode_clip); these numpy versions exist
for callers of the reference API (post-processing, tests of the boundary) and are not used by it.
"""
import numpy as np


def bbox_transform(boxes, deltas, weights=(1.0, 1.0, 1.0, 1.0)):
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]), dtype=deltas.dtype)
    dt = deltas.dtype
    boxes = boxes.astype(dt, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + dt.type(1)
    h = boxes[:, 3] - boxes[:, 1] + dt.type(1)
    cx = boxes[:, 0] + dt.type(0.5) * w
    cy = boxes[:, 1] + dt.type(0.5) * h
    wx, wy, ww, wh = [dt.type(v) for v in weights]
    clip = dt.type(np.log(1000. / 16.))
    dx, dy = deltas[:, 0::4] / wx, deltas[:, 1::4] / wy
    dw, dh = np.minimum(deltas[:, 2::4] / ww, clip), np.minimum(deltas[:, 3::4] / wh, clip)
    pcx, pcy = dx * w[:, None] + cx[:, None], dy * h[:, None] + cy[:, None]
    pw, ph = np.exp(dw) * w[:, None], np.exp(dh) * h[:, None]
    out = np.zeros(deltas.shape, dtype=dt)
    out[:, 0::4] = pcx - dt.type(0.5) * pw
    out[:, 1::4] = pcy - dt.type(0.5) * ph
    out[:, 2::4] = pcx + dt.type(0.5) * pw - dt.type(1)
    out[:, 3::4] = pcy + dt.type(0.5) * ph - dt.type(1)
    return out


def clip_boxes(boxes, im_shape):
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def expand_boxes(boxes, scale):
    """Boxes [n,4] grown about their centres by `scale` (bbox_transform.py:365-381; float64 result like the reference's np.zeros)."""
    half = np.stack([boxes[:, 2] - boxes[:, 0], boxes[:, 3] - boxes[:, 1]], 1) * .5
    ctr = np.stack([boxes[:, 2] + boxes[:, 0], boxes[:, 3] + boxes[:, 1]], 1) * .5
    half = half * scale
    out = np.zeros(boxes.shape)
    out[:, 0:2] = ctr - half
    out[:, 2:4] = ctr + half
    return out


def bbox_overlaps(boxes, query_boxes):
    """IoU matrix [n,k] with the +1 pixel convention (bbox_transform.py:24-42 / bbox.pyx); host helper, not on the inference path."""
    b = np.asarray(boxes, dtype=np.float64)
    q = np.asarray(query_boxes, dtype=np.float64)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + 1
    inter = np.clip(iw, 0, None) * np.clip(ih, 0, None)
    area_b = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    area_q = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    return inter / (area_b[:, None] + area_q[None, :] - inter)
