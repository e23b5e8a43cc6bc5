"""Target program for rocprofv3 --pmc passes over the FPN ROIAlign kernel alone (development aid): N ROIs x 7x7 (or 14x14) on the C1 pyramid,
log-uniform random ROIs, cold (640 MB rewritten before every launch). Usage: python tools/roi_pmc.py <variant> [N] [pooled] [dealt]   (dealt: r13, the ROI -> XCD table of fpn_roi_order)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import gen_rois
from upsnet_amd import ops
from upsnet_amd._lib import lib
variant = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000; ps = int(sys.argv[3]) if len(sys.argv) > 3 else 7
feats = [torch.randn(1, 256, 256 >> l, 512 >> l, device='cuda').contiguous(memory_format=torch.channels_last) for l in range(4)]
flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda')
rois = torch.from_numpy(gen_rois(np.random.default_rng(0), n).astype(np.float32)).cuda()
lib().upsnet_roi_tuning(variant)
order = ops.fpn_roi_order(rois, (1024, 2048)) if (len(sys.argv) > 4 and sys.argv[4] == 'dealt') else None
for _ in range(8):
    flush.add_(1.0)
    out = ops.fpn_roi_align(feats, rois, ps, ps, [0.25, 0.125, 0.0625, 0.03125], order=order)
torch.cuda.synchronize()
