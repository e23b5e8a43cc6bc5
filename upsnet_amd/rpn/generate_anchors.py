"""Base anchor generation (host side, runs once at module construction).

Same interface and values as upsnet/rpn/generate_anchors.py:50-76 (Detectron-style anchors):
``generate_anchors(stride, sizes, aspect_ratios)`` -> float64 [A,4] (x1,y1,x2,y2), ratio-major.
"""
import numpy as np


def _centre(box):
    w = box[2] - box[0] + 1.0
    h = box[3] - box[1] + 1.0
    return w, h, box[0] + 0.5 * (w - 1.0), box[1] + 0.5 * (h - 1.0)


def _boxes_around(ws, hs, cx, cy):
    ws = np.asarray(ws, np.float64).reshape(-1, 1)
    hs = np.asarray(hs, np.float64).reshape(-1, 1)
    return np.concatenate([cx - 0.5 * (ws - 1.0), cy - 0.5 * (hs - 1.0), cx + 0.5 * (ws - 1.0), cy + 0.5 * (hs - 1.0)], axis=1)


def generate_anchors(stride=16, sizes=(32, 64, 128, 256, 512), aspect_ratios=(0.5, 1, 2)):
    scales = np.asarray(sizes, np.float64) / float(stride)
    ratios = np.asarray(aspect_ratios, np.float64)
    cell = np.array([0.0, 0.0, stride - 1.0, stride - 1.0])
    w, h, cx, cy = _centre(cell)
    ws = np.round(np.sqrt(w * h / ratios))   # round-half-even, like the reference's np.round
    hs = np.round(ws * ratios)
    per_ratio = _boxes_around(ws, hs, cx, cy)
    out = []
    for box in per_ratio:
        w, h, cx, cy = _centre(box)
        out.append(_boxes_around(w * scales, h * scales, cx, cy))
    return np.concatenate(out, axis=0)
