"""ROI -> XCD dealing of the FPN ROIAlign launch (csrc/roi_align.hip, fpn_roi_order_kernel, r13): random ROIs (SURVEY 8d's generator) in
generation order vs dealt by image neighbourhood; warm / cold (640 MB read) / cold_dirty (640 MB rewritten), the order kernel timed on its
own, outputs compared bit for bit (development aid; bench.py reports the same figures in roofline.roialign.random_rois)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, torch
from conftest import gen_rois
from upsnet_amd import ops
feats = [torch.randn(1, 256, 256 >> l, 512 >> l, device='cuda').contiguous(memory_format=torch.channels_last) for l in range(4)]
flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda')
sc = [0.25, 0.125, 0.0625, 0.03125]


def t(fn, mode):
    ts = []
    for _ in range(16):
        if mode == 'cold': flush.sum()
        elif mode == 'dirty': flush.add_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000)
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for n, ps in ((1000, 7), (300, 7), (100, 14)):
    rois = torch.from_numpy(gen_rois(np.random.default_rng(0), n).astype(np.float32)).cuda()
    alg = 4 * n * 256 * ps * ps + 20 * n + min(4 * 256 * sum((256 >> l) * (512 >> l) for l in range(4)), 4 * n * 256 * (2 * ps + 1) ** 2)
    order = ops.fpn_roi_order(rois, (1024, 2048))
    assert sorted(order.tolist()) == list(range(n))
    a = ops.fpn_roi_align(feats, rois, ps, ps, sc, order=None)
    b = ops.fpn_roi_align(feats, rois, ps, ps, sc, order=order)
    assert torch.equal(a, b)
    t_order = t(lambda: ops.fpn_roi_order(rois, (1024, 2048)), 'warm')
    for name, o in (('generation order', None), ('dealt per XCD', order)):
        r = {m: t(lambda: ops.fpn_roi_align(feats, rois, ps, ps, sc, order=o), m) for m in ('warm', 'cold', 'dirty')}
        print("N=%4d %2dx%-2d %-18s warm %6.1f us  cold %6.1f us (%.3f of 8 TB/s)  cold_dirty %6.1f us (%.3f)%s" %
              (n, ps, ps, name, r['warm'], r['cold'], alg / r['cold'] / 8e6, r['dirty'], alg / r['dirty'] / 8e6,
               '   [+ order kernel %.1f us]' % t_order if o is not None else ''), flush=True)
