"""Direct MFMA conv vs fused Winograd F(2x2,3x3) on the 3x3 / stride-1 shapes of UPSNet-50 @1024x2048 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd import ops
from upsnet_amd._lib import lib

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

P = [(1, 256 >> l, 512 >> l) for l in range(5)]
shapes = [
    ("FPN P2 3x3", [P[0]], 256, 256), ("FPN P3 3x3", [P[1]], 256, 256), ("FPN P4 3x3", [P[2]], 256, 256), ("FPN P5 3x3", [P[3]], 256, 256),
    ("RPN 3x3 x5 levels", P, 256, 256), ("mask head 3x3 (100 rois)", [(100, 14, 14)], 256, 256), ("mask head 3x3 (143 rois)", [(143, 14, 14)], 256, 256),
    ("res2 3x3", [(1, 256, 512)], 64, 64), ("res3 3x3", [(1, 128, 256)], 128, 128), ("res4 3x3", [(1, 64, 128)], 256, 256), ("res5 3x3", [(1, 32, 64)], 512, 512),
    ("offset conv L0 (4 levels)", P[:4], 256, 18), ("offset conv L1 (4 levels)", P[:4], 128, 18),
]
for name, segs, cin, cout in shapes:
    xs = [torch.randn(n, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    wgt = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    b = torch.randn(cout, device='cuda')
    wd, ldd = ops.pack_conv_weight(wgt)
    td = timeit(lambda: ops.conv2d_nhwc_multi(xs, wd, ldd, b, cout, 3, 1, 1, True))
    ref = ops.conv2d_nhwc_multi(xs, wd, ldd, b, cout, 3, 1, 1, True)
    gf = 2.0 * cout * cin * 9 * sum(n * h * w for n, h, w in segs) / 1e9
    line = "%-28s %6.1f GFLOP | direct %7.1f us (%5.1f TF)" % (name, gf, td, gf / td * 1e3)
    ww, ldw = ops.pack_winograd_weight(wgt)
    for tm in (64, 32):                     # 2x2 tiles per workgroup: 64 = 8 waves, 1 workgroup / CU; 32 = 4 waves, 2 / CU
        lib().upsnet_conv_tuning(tm, 0)
        tw = timeit(lambda: ops.conv2d_winograd_multi(xs, ww, ldw, b, cout, True))
        got = ops.conv2d_winograd_multi(xs, ww, ldw, b, cout, True)
        err = max(float((g - r).abs().max()) for g, r in zip(got, ref))
        line += " | winograd/%d %7.1f us (%5.1f TF-equiv, x%.2f, err %.1e)" % (tm, tw, gf / tw * 1e3, td / tw, err)
    lib().upsnet_conv_tuning(0, 0)
    print(line, flush=True)
