"""Isolated FPN ROIAlign timing at the box-head (1000 x 7x7) and mask-head (100/143 x 14x14) sizes of C1, per kernel variant
(upsnet_roi_tuning: 0 = LDS tap-table kernel, 1 = two register sets, 2 = r03-r07 kernel), WARM (features resident in the 256 MiB
Infinity Cache: 178 MB re-read every iteration) and COLD (a 640 MB buffer is rewritten between calls, as in the model, where the
RPN / head activations of the image pass through the cache between the FPN and the ROIAlign). Development aid."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from conftest import gen_rois
from upsnet_amd import ops
from upsnet_amd._lib import lib
feats = [torch.randn(1, 256, 256 >> l, 512 >> l, device='cuda').contiguous(memory_format=torch.channels_last) for l in range(4)]
flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda')
sc = [0.25, 0.125, 0.0625, 0.03125]
for n, ps in ((1000, 7), (100, 14), (143, 14), (300, 7)):
    rois = torch.from_numpy(gen_rois(np.random.default_rng(0), n).astype(np.float32)).cuda()
    feat_bytes = 4 * 256 * sum((256 >> l) * (512 >> l) for l in range(4))
    alg = 4 * n * 256 * ps * ps + 20 * n + min(feat_bytes, 4 * n * 256 * (2 * ps + 1) ** 2)
    ref = None
    for variant in (2, 0, 1):
        lib().upsnet_roi_tuning(variant)
        for _ in range(3): out = ops.fpn_roi_align(feats, rois, ps, ps, sc)
        ref = out if ref is None else ref
        same = bool(torch.equal(out, ref))
        res = []
        for cold in (False, True):
            ts = []
            for _ in range(12):
                if cold:
                    flush.add_(1.0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); out = ops.fpn_roi_align(feats, rois, ps, ps, sc); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1000)
            ts = sorted(ts[2:])
            res.append(ts[len(ts) // 2])
        print("N=%4d %2dx%-2d variant %d: warm %6.1f us, cold %6.1f us (alg %.1f MB -> cold %.2f TB/s = %.1f %% of 8 TB/s) same bits %s" %
              (n, ps, ps, variant, res[0], res[1], alg / 1e6, alg / res[1] / 1e6, alg / res[1] / 1e6 / 8 * 100, same), flush=True)
lib().upsnet_roi_tuning(0)
