"""Isolated FPN ROIAlign timing at the box-head (1000 x 7x7) and mask-head (143 x 14x14) sizes of C1 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import gen_rois
from upsnet_amd import ops
feats = [torch.randn(1, 256, 256 >> l, 512 >> l, device='cuda').contiguous(memory_format=torch.channels_last) for l in range(4)]
for n, ps in ((1000, 7), (143, 14), (300, 7)):
    rois = torch.from_numpy(gen_rois(np.random.default_rng(0), n).astype(np.float32)).cuda()
    for _ in range(3): out = ops.fpn_roi_align(feats, rois, ps, ps, [0.25, 0.125, 0.0625, 0.03125])
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): out = ops.fpn_roi_align(feats, rois, ps, ps, [0.25, 0.125, 0.0625, 0.03125])
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1000
    feat_bytes = 4 * 256 * sum((256 >> l) * (512 >> l) for l in range(4))
    alg = 4 * n * 256 * ps * ps + 20 * n + min(feat_bytes, 4 * n * 256 * (2 * ps + 1) ** 2)
    print("N=%d %dx%d: %.1f us, algorithmic %.1f MB -> %.2f TB/s = %.1f %% of 8 TB/s" % (n, ps, ps, us, alg / 1e6, alg / us / 1e6, alg / us / 1e6 / 8 * 100))
