"""Split-K F(4x4,3x3) (csrc/conv_wino36.hip, conv_wino36_f32_kernel<true> + reduce, r13) on the single maps that are too small for the unsplit
F(4x4) form (fewer workgroups than CUs), against what hipconv picks today (F(2x2), unsplit or its own split-K); error of each against
float64 at rtol = atol = 1e-4 (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from upsnet_amd import ops
from upsnet_amd.models import hipconv
from gputime import gpu_time

shapes = [("c1 res3 conv2 128->128", (1, 128, 256), 128, 128), ("c1 res4 conv2 256->256", (1, 64, 128), 256, 256), ("c1 FPN P4 256->256", (1, 64, 128), 256, 256),
          ("c1 res5 conv2 512->512", (1, 32, 64), 512, 512), ("c1 FPN P5 256->256", (1, 32, 64), 256, 256),
          ("c3 FPN P3 256->256", (1, 100, 168), 256, 256), ("c3 FPN P4 256->256", (1, 50, 84), 256, 256), ("c3 res2 conv2 64->64", (1, 200, 336), 64, 64)]
for name, (n, h, w), cin, cout in shapes:
    torch.manual_seed(0)
    m = torch.nn.Conv2d(cin, cout, 3, padding=1).cuda()
    x = torch.randn(n, cin, h, w, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m.weight.copy_(torch.randn(cout, cin, 3, 3, device='cuda') * (2.0 / (cin * 9)) ** 0.5)
        ref = F.relu(F.conv2d(x.double(), m.weight.double(), m.bias.double(), padding=1))
        lim = 1e-4 + 1e-4 * ref.abs()
        was, hipconv.WINO36_SPLITK = getattr(hipconv, 'WINO36_SPLITK', False), False
        hipconv.TRACE = []
        y0 = hipconv.conv(m, x, relu=True)
        form = hipconv.TRACE[-1]['form']
        hipconv.TRACE = None
        t0 = gpu_time(lambda: hipconv.conv(m, x, relu=True), n=8)
        hipconv.WINO36_SPLITK = was
        e0 = float(((y0.double() - ref).abs() / lim).max())
        w36, ld36 = ops.pack_winograd36_weight(m.weight)
        line = "%-26s %-22s %6.1f us (err/bound %.3f) | F(4x4) unsplit %6.1f |" % (name, form, t0, e0, gpu_time(lambda: ops.conv2d_winograd36_multi([x], w36, ld36, m.bias, cout, True), n=8))
        for ks in (2, 3, 4, 6, 8):
            if cin // 16 < ks: continue
            y = ops.conv2d_winograd36_splitk(x, w36, ld36, m.bias, cout, ks, relu=True)
            e = float(((y.double() - ref).abs() / lim).max())
            line += " x%d %6.1f (%.3f)" % (ks, gpu_time(lambda: ops.conv2d_winograd36_splitk(x, w36, ld36, m.bias, cout, ks, relu=True), n=8), e)
    print(line, flush=True)
