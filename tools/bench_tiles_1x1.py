"""Tile forms of the direct kernel on the HBM-bound short-K 1x1 layers (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd import ops
from upsnet_amd._lib import lib

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

names = {5: '64x64', 4: '64x128', 2: '128x64', 1: '128x128'}
for name, (n, h, w), cin, cout, res in [("res2 conv3 64->256 +res", (1, 256, 512), 64, 256, True), ("res2 conv1 256->64", (1, 256, 512), 256, 64, False),
                                        ("res3 conv3 128->512 +res", (1, 128, 256), 128, 512, True), ("res4 conv3 256->1024 +res", (1, 64, 128), 256, 1024, True)]:
    x = torch.randn(n, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    b = torch.randn(cout, device='cuda')
    r = torch.randn(n, cout, h, w, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    wp, ldw = ops.pack_conv_weight(wgt)
    mb = 4e-6 * n * h * w * (cin + cout * (2 if res else 1))
    line = "%-28s %5.0f MB" % (name, mb)
    for tile in (5, 4, 2, 1):
        lib().upsnet_conv_tuning(0, tile)
        t = timeit(lambda: ops.conv2d_nhwc(x, wp, ldw, b, cout, 1, 1, 0, relu=True, residual=r))
        line += " | %s %6.1f us (%4.2f TB/s)" % (names[tile], t, mb / t)
    lib().upsnet_conv_tuning(0, 0)
    print(line, flush=True)
