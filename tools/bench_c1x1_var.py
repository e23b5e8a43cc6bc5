import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from upsnet_amd import ops
from upsnet_amd._lib import lib
from gputime import gpu_time as timeit

shapes = [("res4 conv1 1024->256 50x84", 50, 84, 1024, 256), ("res4 conv1 1024->256 64x128", 64, 128, 1024, 256), ("res3 conv1 512->128 128x256", 128, 256, 512, 128),
          ("res5 conv1 2048->512 32x64", 32, 64, 2048, 512), ("res5 conv1 2048->512 25x42", 25, 42, 2048, 512), ("res3 conv1 512->128 100x168", 100, 168, 512, 128)]
for name, H, W, cin, cout in shapes:
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    b = torch.randn(cout, device='cuda')
    wf = ops.pack_conv1x1_weight(wgt)
    line = "%-32s" % name
    ref = None
    for tune in (64, 4064, 5064, 6064, 7064, 8064):
        lib().upsnet_conv1x1_tuning(tune)
        y = ops.conv1x1_frag(x, wf, b, cout, 1, relu=True)
        ref = y if ref is None else ref
        line += " v%d %5.1f |" % (tune // 1000, timeit(lambda: ops.conv1x1_frag(x, wf, b, cout, 1, relu=True)))
    lib().upsnet_conv1x1_tuning(0)
    print(line + "  peak %.1f us" % (2.0 * cin * cout * H * W / 157.3e6), flush=True)
