"""Time the 1x1 layers of the bf16 mode outside the fused identity blocks (first bottleneck of each stage, FPN laterals) at the shapes
of the 1024x2048 workload: conv_bf16_kernel against csrc/conv1x1_wreg_bf16.hip. GPU only."""
import sys
import torch

sys.path.insert(0, '.')
from upsnet_amd import ops  # noqa: E402
from upsnet_amd._lib import lib  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    bf = torch.bfloat16
    cases = [  # name, Cin, Cout, H, W (input), stride, shortcut, out dtype
        ('res2 projection', 64, 256, 256, 512, 1, None, bf), ('res2 conv3 + shortcut', 64, 256, 256, 512, 1, 'same', bf),
        ('res3 conv1 /2', 256, 128, 256, 512, 2, None, bf), ('res3 projection /2', 256, 512, 256, 512, 2, None, bf),
        ('res3 conv3 + shortcut', 128, 512, 128, 256, 1, 'same', bf),
        ('res4 conv1 /2', 512, 256, 128, 256, 2, None, bf), ('res4 projection /2', 512, 1024, 128, 256, 2, None, bf),
        ('res4 conv3 + shortcut', 256, 1024, 64, 128, 1, 'same', bf),
        ('res5 conv1 /2', 1024, 512, 64, 128, 2, None, bf), ('res5 projection /2', 1024, 2048, 64, 128, 2, None, bf),
        ('res5 conv3 + shortcut', 512, 2048, 32, 64, 1, 'same', bf),
        ('lateral P5', 2048, 256, 32, 64, 1, None, torch.float32), ('lateral P4 + up', 1024, 256, 64, 128, 1, 'up32', torch.float32),
        ('lateral P3 + up', 512, 256, 128, 256, 1, 'up32', bf), ('lateral P2 + up', 256, 256, 256, 512, 1, 'up16', bf),
    ]
    tot = [0.0, 0.0]
    for name, cin, cout, H, W, st, sc, od in cases:
        x = torch.randn(1, cin, H, W, device='cuda').to(bf).contiguous(memory_format=torch.channels_last)
        w = (torch.randn(cout, cin, 1, 1) / cin ** 0.5).cuda()
        b = torch.randn(cout).cuda()
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        r, up = None, False
        if sc == 'same':
            r = torch.randn(1, cout, Ho, Wo, device='cuda').to(bf).contiguous(memory_format=torch.channels_last)
        elif sc is not None:
            up = True
            r = torch.randn(1, cout, Ho // 2, Wo // 2, device='cuda').contiguous(memory_format=torch.channels_last)
            r = r.to(bf) if sc == 'up16' else r
        hi, _, ldw = ops.pack_conv_weight_bf16(w, split=False)
        run = lambda: ops.conv2d_nhwc_bf16_multi([x], hi, None, ldw, b, cout, 1, st, 0, relu=True, residuals=None if r is None else [r],
                                                 residual_up=up, out_dtype=od)
        t = []
        for en in (0, 2):
            lib().upsnet_conv1x1_bf16_tuning(en)
            t.append(timeit(run))
        lib().upsnet_conv1x1_bf16_tuning(1)
        esz = 2 if od == bf else 4
        byt = 2.0 * cin * Ho * Wo + esz * cout * Ho * Wo + (0 if r is None else r.numel() * r.element_size()) + 2.0 * cin * cout
        tot[0] += t[0]; tot[1] += t[1]
        print("%-24s %4d -> %4d  %3dx%3d  conv_bf16 %6.1f us   wreg %6.1f us  (%5.0f GB/s algorithmic)" % (name, cin, cout, Ho, Wo, t[0], t[1], byt / t[1] * 1e-3),
              flush=True)
    print("sum: conv_bf16 %.1f us   wreg %.1f us" % tuple(tot))


if __name__ == '__main__':
    main()
