"""The per-block layers of UPSNet-101-DCN at 800x1333 (BASELINE configs[3]) in isolation: the lean 1x1 GEMM kernel vs the general kernel
with split-K, and the fused deformable convolution at several split-K factors (development aid, r10)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from upsnet_amd import ops

from gputime import gpu_time as timeit

shapes = [("res4 conv1 1024->256", 50, 84, 1024, 256, False), ("res4 conv3 256->1024 +res", 50, 84, 256, 1024, True),
          ("res3 conv1 512->128", 100, 168, 512, 128, False), ("res3 conv3 128->512 +res", 100, 168, 128, 512, True),
          ("res5 conv1 2048->512", 25, 42, 2048, 512, False), ("res5 conv3 512->2048 +res", 25, 42, 512, 2048, True),
          ("res4 conv1 1024->256 (C1 64x128)", 64, 128, 1024, 256, False), ("res4 conv3 256->1024 +res (C1)", 64, 128, 256, 1024, True)]
for name, H, W, cin, cout, res in shapes:
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    b = torch.randn(cout, device='cuda')
    wp, ldw = ops.pack_conv_weight(wgt)
    wf = ops.pack_conv1x1_weight(wgt)
    r = torch.randn(1, cout, H, W, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    line = "%-34s lean %6.1f us | general %6.1f |" % (name, timeit(lambda: ops.conv1x1_frag(x, wf, b, cout, 1, relu=True, residual=r)),
                                                        timeit(lambda: ops.conv2d_nhwc(x, wp, ldw, b, cout, 1, 1, 0, relu=True, residual=r)))
    for ks in (2, 3, 4, 6, 8):
        if (cin // 32 + ks - 1) // ks * (ks - 1) >= cin // 32: continue
        line += " x%d %6.1f" % (ks, timeit(lambda: ops.conv2d_nhwc_splitk(x, wp, ldw, b, cout, 1, 1, 0, ks, relu=True, residual=r)))
    flops = 2.0 * cin * cout * H * W
    print(line + "   (%.1f us at the fp32 MFMA peak)" % (flops / 157.3e6), flush=True)

real_ksplit = ops.dcn_ksplit
for name, H, W, c in (("res4 dcn 256->256", 50, 84, 256), ("res3 dcn 128->128", 100, 168, 128), ("res5 dcn 512->512", 25, 42, 512)):
    x = torch.randn(1, c, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    off = (torch.randn(1, 18, H, W, device='cuda')).contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(c, c, 3, 3, device='cuda') / (9 * c) ** 0.5
    wpk = ops.pack_dcn_weight(wgt)
    line = "%-34s auto(x%d) %6.1f us |" % (name, real_ksplit([torch.empty(1, c, H, W)], c, c, 9),
                                            timeit(lambda: ops.deform_conv_fused([x], [off], wpk, None, c, c, (3, 3), (1, 1), (1, 1), (1, 1), relu=True)))
    for ks in (1, 2, 3, 4, 6, 8):
        ops.dcn_ksplit = lambda *a, _k=ks: _k
        try:
            line += " x%d %6.1f" % (ks, timeit(lambda: ops.deform_conv_fused([x], [off], wpk, None, c, c, (3, 3), (1, 1), (1, 1), (1, 1), relu=True)))
        except Exception as e:
            line += " x%d  n/a " % ks
    ops.dcn_ksplit = real_ksplit
    print(line + "   (%.1f us at the fp32 MFMA peak)" % (2.0 * 9 * c * c * H * W / 157.3e6), flush=True)
