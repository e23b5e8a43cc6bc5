"""The small-tile 1x1 kernel (csrc/conv1x1_ksw.hip: 16x16x4 MFMA fragments, K split over the waves) against the 64-pixel kernel
(csrc/conv1x1.hip) on the 1x1 layers of UPSNet-101-DCN at 800x1333 (BASELINE configs[3]) and of the headline workload, all four tiles;
every result is checked against float64 at rtol = atol = 1e-4 on the first shape of each group (development aid, r13)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from upsnet_amd import ops

from gputime import gpu_time as timeit

TILES = [((16, 64), 0), ((32, 32), 0), ((32, 64), 0), ((64, 64), 0), ((16, 256), 1), ((32, 128), 1), ((32, 256), 1)]
shapes = [("c3 res4 conv1 1024->256", 50, 84, 1024, 256, False, 1), ("c3 res4 conv3 256->1024 +res", 50, 84, 256, 1024, True, 1),
          ("c3 res3 conv1 512->128", 100, 168, 512, 128, False, 1), ("c3 res3 conv3 128->512 +res", 100, 168, 128, 512, True, 1),
          ("c3 res5 conv1 2048->512", 25, 42, 2048, 512, False, 1), ("c3 res5 conv3 512->2048 +res", 25, 42, 512, 2048, True, 1),
          ("c3 res2 conv1 256->64", 200, 336, 256, 64, False, 1), ("c3 res2 conv3 64->256 +res", 200, 336, 64, 256, True, 1),
          ("c3 lateral 2048->256", 25, 42, 2048, 256, False, 1), ("c3 res4 first conv1 /2 512->256", 100, 168, 512, 256, False, 2),
          ("c1 res4 conv1 1024->256", 64, 128, 1024, 256, False, 1), ("c1 res4 conv3 256->1024 +res", 64, 128, 256, 1024, True, 1),
          ("c1 res5 conv1 2048->512", 32, 64, 2048, 512, False, 1), ("c1 res5 conv3 512->2048 +res", 32, 64, 512, 2048, True, 1),
          ("c1 res3 conv1 512->128", 128, 256, 512, 128, False, 1), ("c1 res3 conv3 128->512 +res", 128, 256, 128, 512, True, 1),
          ("c1 lateral P2 256->256 +res", 256, 512, 256, 256, True, 1)]
if len(sys.argv) > 1:
    shapes = [s for s in shapes if sys.argv[1] in s[0]]
for name, H, W, cin, cout, res, st in shapes:
    torch.manual_seed(cin + cout)
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    wgt = torch.randn(cout, cin, 1, 1, device='cuda') / cin ** 0.5
    b = torch.randn(cout, device='cuda')
    wf = ops.pack_conv1x1_weight(wgt)
    wk = ops.pack_conv1x1_ksw_weight(wgt)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    r = torch.randn(1, cout, Ho, Wo, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    ref = F.conv2d(x.double(), wgt.double(), b.double(), stride=st) + (r.double() if res else 0)
    ref = ref.clamp_min(0)
    line = "%-34s frag64 %6.1f us |" % (name, timeit(lambda: ops.conv1x1_frag(x, wf, b, cout, st, relu=True, residual=r)))
    for t, sn in TILES:
        y = ops.conv1x1_ksw(x, wk, b, cout, t, stride=st, relu=True, residual=r, split_n=sn)
        worst = float(((y.double() - ref).abs() / (1e-4 + 1e-4 * ref.abs())).max())
        line += " %s%dx%d %5.1f (%.3f)" % ('n' if sn else 'k', t[0], t[1], timeit(lambda: ops.conv1x1_ksw(x, wk, b, cout, t, stride=st, relu=True, residual=r, split_n=sn)), worst)
    flops = 2.0 * cin * cout * Ho * Wo
    print(line + "   (%.1f us at the fp32 MFMA peak; in brackets: worst error / the 1e-4 bound)" % (flops / 157.3e6), flush=True)

# ---- the same layers through models/hipconv.py: the form it picks with and without the small-tile kernel
from upsnet_amd.models import hipconv
print("\nthrough hipconv.conv (UPSNET_CONV1X1_KSW=0 vs 1):")
for name, H, W, cin, cout, res, st in shapes + [("c1 lateral P5 2048->256", 32, 64, 2048, 256, False, 1), ("c3 res5 first conv1 /2 1024->512", 50, 84, 1024, 512, False, 2),
                                                 ("c4 (1024x2048 R101) res5 conv1 2048->512", 32, 64, 2048, 512, False, 1)]:
    torch.manual_seed(cin + cout)
    m = torch.nn.Conv2d(cin, cout, 1, stride=st).cuda()
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
    r = torch.randn(1, cout, Ho, Wo, device='cuda').contiguous(memory_format=torch.channels_last) if res else None
    out = []
    with torch.no_grad():
        for on in (False, True):
            hipconv.KSW = on
            hipconv.TRACE = []
            hipconv.conv(m, x, relu=True, residual=r)
            form = hipconv.TRACE[-1]['form']
            hipconv.TRACE = None
            out.append("%-22s %6.1f us" % (form, timeit(lambda: hipconv.conv(m, x, relu=True, residual=r))))
    print("%-42s %s | %s" % (name, out[0], out[1]), flush=True)
