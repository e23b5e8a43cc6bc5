"""Where the fixed time of a launch goes (r10): the 1x1 GEMM kernel and the 32x64 Winograd form on the res4 map of 1024x2048 (64 x 128 pixels,
one workgroup per CU) with the K walk shortened step by step -- the intercept of time against K is what a launch costs besides its MFMAs.
Graph-replay-timed (period of back-to-back launches = duration + boundary); a trivial kernel gives the floor."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from upsnet_amd import ops
from gputime import gpu_time
torch.set_grad_enabled(False)

H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 128)
z = torch.zeros(64, device='cuda')
print("trivial kernel (64-element fill): %5.1f us" % gpu_time(lambda: z.zero_(), n=40), flush=True)
for cout in (256, 1024):
    for full in (False, True):
        row = []
        for cin in (32, 64, 128, 256, 512, 1024):
            if cout == 1024 and cin > 256:
                continue
            x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
            w = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
            b = torch.randn(cout, device='cuda') if full else None
            r = torch.randn(1, cout, H, W, device='cuda').contiguous(memory_format=torch.channels_last) if full else None
            pk = ops.pack_conv1x1_weight(w)
            row.append((cin, gpu_time(lambda: ops.conv1x1_frag(x, pk, b, cout, 1, relu=full, residual=r), n=20)))
        print("conv1x1 ->%4d %-18s " % (cout, "bias+residual+relu" if full else "plain") + "  ".join("K=%d %5.1f" % t for t in row), flush=True)
from upsnet_amd.models import hipconv
import torch.nn as nn
for cout in (256,):
    row = []
    for cin in (32, 64, 128, 256):
        m = nn.Conv2d(cin, cout, 3, padding=1).cuda()
        x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
        wp, ldw = hipconv._winograd_plan(m)
        row.append((cin, gpu_time(lambda: ops.conv2d_winograd_multi([x], wp, ldw, m.bias, cout, relu=True), n=20), ops.last_kernel_form()))
    print("winograd 3x3 ->%4d " % cout + "  ".join("K=%d %5.1f (%s)" % t for t in row), flush=True)
