import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tools'))
import torch
from upsnet_amd import ops
from gputime import gpu_time
for H, W, cin, cout in ((8, 8, 1024, 256), (8, 8, 64, 64), (16, 64, 1024, 256), (32, 64, 1024, 256), (50, 84, 1024, 256), (50, 84, 512, 256), (50, 84, 256, 256), (50, 84, 128, 256), (50, 84, 64, 256)):
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    wf = ops.pack_conv1x1_weight(wgt)
    print("%dx%d %d->%d: %.1f us" % (H, W, cin, cout, gpu_time(lambda: ops.conv1x1_frag(x, wf, None, cout, 1, relu=True), n=30)), flush=True)
y = torch.zeros(1024, device='cuda')
print("tiny add_: %.1f us" % gpu_time(lambda: y.add_(1.0), n=50))
