"""The small-tile 3x3 -> <= 32 channel kernel (csrc/conv1x1_ksw.hip: conv3x3_ksw_f32_kernel) against what models/hipconv.py picks without it
(Winograd 32-channel form / general kernel with split-K) on the offset predictors of the deformable bottlenecks of UPSNet-101-DCN at
800x1333 and 1024x2048; checked against float64 at rtol = atol = 1e-4 (development aid, r13)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from upsnet_amd import ops
from upsnet_amd.models import hipconv
from gputime import gpu_time as timeit

for name, H, W, cin in [("c3 res3 128->18", 100, 168, 128), ("c3 res4 256->18", 50, 84, 256), ("c3 res5 512->18", 25, 42, 512),
                        ("c4 res3 128->18", 128, 256, 128), ("c4 res4 256->18", 64, 128, 256), ("c4 res5 512->18", 32, 64, 512),
                        ("fcn P2 256->18", 256, 512, 256), ("fcn P3 256->18", 128, 256, 256)]:
    torch.manual_seed(cin)
    m = torch.nn.Conv2d(cin, 18, 3, padding=1).cuda()
    x = torch.randn(1, cin, H, W, device='cuda').relu_().contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        m.weight.mul_(0.5)
        ref = F.conv2d(x.double(), m.weight.double(), m.bias.double(), padding=1)
        wk = ops.pack_conv3x3_ksw_weight(m.weight)
        y = ops.conv3x3_ksw(x, wk, m.bias, 18)
        worst = float(((y.double() - ref).abs() / (1e-4 + 1e-4 * ref.abs())).max())
        was, hipconv.KSW3 = getattr(hipconv, 'KSW3', False), False
        hipconv.TRACE = []
        y0 = hipconv.conv(m, x)
        form = hipconv.TRACE[-1]['form']
        hipconv.TRACE = None
        worst0 = float(((y0.double() - ref).abs() / (1e-4 + 1e-4 * ref.abs())).max())
        t0 = timeit(lambda: hipconv.conv(m, x))
        hipconv.KSW3 = was
        t1 = timeit(lambda: ops.conv3x3_ksw(x, wk, m.bias, 18))
    print("%-18s %-22s %6.1f us (%.3f) | ksw 16x32 %6.1f us (%.3f)   (%.1f us at the fp32 MFMA peak for 32 columns)" %
          (name, form, t0, worst0, t1, worst, 2.0 * 32 * cin * 9 * H * W / 157.3e6), flush=True)
