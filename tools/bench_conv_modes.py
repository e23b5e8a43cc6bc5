"""fp32 direct / fp32 Winograd / bf16x3 / bf16 instances of the convolution on representative shapes of UPSNet-50 @1024x2048."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd import ops

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
    ("FPN P2 3x3", [P[0]], 256, 256, 3, 1), ("RPN 3x3 x5 levels", P, 256, 256, 3, 1), ("mask head 3x3 (100 rois)", [(100, 14, 14)], 256, 256, 3, 1),
    ("res2 3x3", [(1, 256, 512)], 64, 64, 3, 1), ("res3 3x3", [(1, 128, 256)], 128, 128, 3, 1), ("res4 3x3", [(1, 64, 128)], 256, 256, 3, 1),
    ("res5 3x3", [(1, 32, 64)], 512, 512, 3, 1),
    ("res2 1x1 64->256", [(1, 256, 512)], 64, 256, 1, 1), ("res2 1x1 256->64", [(1, 256, 512)], 256, 64, 1, 1),
    ("res3 1x1 128->512", [(1, 128, 256)], 128, 512, 1, 1), ("res4 1x1 256->1024", [(1, 64, 128)], 256, 1024, 1, 1),
    ("res4 1x1 1024->256", [(1, 64, 128)], 1024, 256, 1, 1), ("res5 1x1 2048->512", [(1, 32, 64)], 2048, 512, 1, 1),
    ("res4 1x1 s2 512->1024", [(1, 128, 256)], 512, 1024, 1, 2),
]
only = os.environ.get('ONLY')
for name, segs, cin, cout, k, st in shapes:
    if only and only not in name:
        continue
    xs = [torch.randn(n, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    wgt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    b = torch.randn(cout, device='cuda')
    wd, ldd = ops.pack_conv_weight(wgt)
    hi, lo, ldb = ops.pack_conv_weight_bf16(wgt, split=True)
    td = timeit(lambda: ops.conv2d_nhwc_multi(xs, wd, ldd, b, cout, k, st, k // 2, True))
    t3 = timeit(lambda: ops.conv2d_nhwc_bf16_multi(xs, hi, lo, ldb, b, cout, k, st, k // 2, True))
    t1 = timeit(lambda: ops.conv2d_nhwc_bf16_multi(xs, hi, None, ldb, b, cout, k, st, k // 2, True))
    tw = float('nan')
    if k == 3 and st == 1:
        ww, ldw = ops.pack_winograd_weight(wgt)
        tw = timeit(lambda: ops.conv2d_winograd_multi(xs, ww, ldw, b, cout, True))
    npix = sum(n * ((h - 1) // st + 1) * ((w - 1) // st + 1) for n, h, w in segs)
    gf = 2.0 * cout * cin * k * k * npix / 1e9
    print("%-26s %6.1f GF | fp32 %7.1f us | winograd %7.1f | bf16x3 %7.1f (x%.2f) | bf16 %7.1f (x%.2f)" % (name, gf, td, tw, t3, td / t3, t1, td / t1), flush=True)
