"""Mask head 3x3 layer (N ROIs x 256 x 14 x 14) on the Winograd kernel: the 32 x 64 form, the 32 x 32 form, and both in one launch
(conv_wino16_tail_f32_kernel) for several main / tail splits; graph-replay-timed (r10, profiles/HISTORY.md)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch, torch.nn as nn
from upsnet_amd import ops
from upsnet_amd.models import hipconv
from gputime import gpu_time
torch.set_grad_enabled(False)
m = nn.Conv2d(256, 256, 3, padding=1).cuda()
for n in (100, 104, 83, 64):
    x = torch.randn(n, 256, 14, 14, device='cuda').contiguous(memory_format=torch.channels_last)
    wp, ldw = hipconv._winograd_plan(m)
    wp32, ldw32 = hipconv._winograd_plan(m, tn32=True)
    t0 = gpu_time(lambda: ops.conv2d_winograd_multi([x], wp, ldw, m.bias, 256, relu=True), n=10); f0 = ops.last_kernel_form()
    t1 = gpu_time(lambda: ops.conv2d_winograd_multi([x], wp32, ldw32, m.bias, 256, relu=True, tn32=True), n=10); f1 = ops.last_kernel_form()
    print(n, "rois: %s %.1f us   %s %.1f us" % (f0, t0, f1, t1), flush=True)
x = torch.randn(100, 256, 14, 14, device='cuda').contiguous(memory_format=torch.channels_last)
for nm in (83, 80, 78, 74):
    t = gpu_time(lambda: ops.conv2d_winograd_tail(x, wp, ldw, wp32, ldw32, m.bias, 256, nm, relu=True), n=10)
    print("fused tail, n_main %d: %s %.1f us" % (nm, ops.last_kernel_form(), t), flush=True)
