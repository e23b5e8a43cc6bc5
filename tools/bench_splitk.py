"""Unsplit vs split-K on the small-map conv shapes of UPSNet-50 @1024x2048 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd import ops

def timeit(fn, n=8):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

shapes = [("res4 3x3 256", 64, 128, 256, 256, 3, 1, False), ("res4 1x1 256->1024 res", 64, 128, 256, 1024, 1, 1, True), ("res4 1x1 1024->256", 64, 128, 1024, 256, 1, 1, False),
          ("res5 3x3 512", 32, 64, 512, 512, 3, 1, False), ("res5 1x1 512->2048 res", 32, 64, 512, 2048, 1, 1, True), ("res5 1x1 2048->512", 32, 64, 2048, 512, 1, 1, False),
          ("res5 1x1 s2 1024->2048", 64, 128, 1024, 2048, 1, 2, False), ("res5 1x1 s2 1024->512", 64, 128, 1024, 512, 1, 2, False),
          ("FPN P4 3x3", 64, 128, 256, 256, 3, 1, False), ("FPN P5 3x3", 32, 64, 256, 256, 3, 1, False), ("FPN P5 lateral 2048->256", 32, 64, 2048, 256, 1, 1, False),
          ("res3 3x3 128", 128, 256, 128, 128, 3, 1, False), ("res3 1x1 512->128", 128, 256, 512, 128, 1, 1, False)]
for name, H, W, cin, cout, k, st, res in shapes:
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    b = torch.randn(cout, device='cuda')
    wp, ldw = ops.pack_conv_weight(wgt)
    r = torch.randn(1, cout, (H - 1) // st + 1, (W - 1) // st + 1, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    t1 = timeit(lambda: ops.conv2d_nhwc(x, wp, ldw, b, cout, k, st, k // 2, relu=True, residual=r))
    line = "%-26s unsplit %6.1f us |" % (name, t1)
    for ks in (2, 3, 4, 8):
        if (k * k * cin // 32 + ks - 1) // ks * (ks - 1) >= k * k * cin // 32: continue
        t = timeit(lambda: ops.conv2d_nhwc_splitk(x, wp, ldw, b, cout, k, st, k // 2, ks, relu=True, residual=r))
        line += " x%d %6.1f" % (ks, t)
    print(line, flush=True)
