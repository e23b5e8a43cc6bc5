"""Target program for rocprofv3 --pmc passes over the F(4x4) Winograd kernel alone (development aid, r13): one 1 x Cin x 256 x 512 map -> Cout
channels, 6 launches. Usage: python tools/wino36_pmc.py <Cin> <Cout>. What is being asked: the kernel fetches ~2x its
algorithmic bytes at the fabric -- the input patches (re-read by the n-tile siblings of an m-tile) or the transformed weights (36 Cin Cout
floats = 9.4 MB at 256 -> 256, more than an XCD's 4 MB L2, walked once per workgroup round)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd import ops
cin, cout = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(1, cin, 256, 512, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
w = torch.randn(cout, cin, 3, 3, device='cuda') * (2.0 / (9 * cin)) ** 0.5
b = torch.randn(cout, device='cuda')
wp, ldw = ops.pack_winograd36_weight(w)
flush = torch.empty(160 << 20, dtype=torch.float32, device='cuda')
for _ in range(6):
    flush.add_(1.0)
    y = ops.conv2d_winograd36_multi([x], wp, ldw, b, cout, True)
torch.cuda.synchronize()
