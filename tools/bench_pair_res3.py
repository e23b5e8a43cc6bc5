"""res3 block boundary at 1024x2048 (128 x 256 map): conv3 128->512 + shortcut + ReLU and conv1 512->128 + ReLU as two launches vs the pair
kernel (r10), graph-replay-timed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from upsnet_amd import ops
from gputime import gpu_time
for name, H, W, c0, c1 in (("res3 1024x2048", 128, 256, 128, 512), ("res3 800x1333", 100, 168, 128, 512), ("res2 1024x2048", 256, 512, 64, 256), ("res4 1024x2048", 64, 128, 256, 1024), ("res4 800x1333", 50, 84, 256, 1024)):
    x = torch.randn(1, c0, H, W, device='cuda').relu().contiguous(memory_format=torch.channels_last)
    sc = torch.randn(1, c1, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w3 = torch.randn(c1, c0, 1, 1, device='cuda') / c0 ** 0.5
    w1 = torch.randn(c0, c1, 1, 1, device='cuda') / c1 ** 0.5
    b3, b1 = torch.randn(c1, device='cuda'), torch.randn(c0, device='cuda')
    p3, p1 = ops.pack_conv1x1_weight(w3), ops.pack_conv1x1_weight(w1)
    def two():
        s1 = ops.conv1x1_frag(x, p3, b3, c1, 1, relu=True, residual=sc)
        return ops.conv1x1_frag(s1, p1, b1, c0, 1, relu=True)
    t2 = gpu_time(two, n=10)
    tp = gpu_time(lambda: ops.conv1x1_pair(x, sc, p3, b3, c1, p1, b1, c0), n=10)
    print("%-16s two launches %6.1f us   pair %6.1f us" % (name, t2, tp), flush=True)
