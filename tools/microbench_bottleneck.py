"""Time one identity bottleneck of the bf16 mode at the four backbone shapes of the 1024x2048 workload: the one-launch kernel
(csrc/bottleneck_bf16.hip) against the three launches it replaces. GPU only."""
import sys
import torch

sys.path.insert(0, '.')
from upsnet_amd import ops  # noqa: E402


def timeit(fn, n=30):
    for _ in range(5):
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
    for cm, H, W in [(64, 256, 512), (128, 128, 256), (256, 64, 128), (512, 32, 64)]:
        c = 4 * cm
        g = torch.Generator().manual_seed(cm)
        w1 = (torch.randn(cm, c, 1, 1, generator=g) / c ** 0.5).cuda()
        w2 = (torch.randn(cm, cm, 3, 3, generator=g) / (9 * cm) ** 0.5).cuda()
        w3 = (torch.randn(c, cm, 1, 1, generator=g) / cm ** 0.5).cuda()
        b1, b2, b3 = torch.zeros(cm).cuda(), torch.zeros(cm).cuda(), torch.zeros(c).cuda()
        x = torch.randn(1, c, H, W, device='cuda').to(bf).contiguous(memory_format=torch.channels_last)
        pack = ops.pack_bottleneck_bf16(w1, w2, w3, b1, b2, b3)
        p1, p2, p3 = (ops.pack_conv_weight_bf16(w, split=False) for w in (w1, w2, w3))

        def sep():
            t1 = ops.conv2d_nhwc_bf16_multi([x], p1[0], None, p1[2], b1, cm, 1, 1, 0, relu=True, out_dtype=bf)[0]
            t2 = ops.conv2d_nhwc_bf16_multi([t1], p2[0], None, p2[2], b2, cm, 3, 1, 1, relu=True, out_dtype=bf)[0]
            return ops.conv2d_nhwc_bf16_multi([t2], p3[0], None, p3[2], b3, c, 1, 1, 0, relu=True, residuals=[x], out_dtype=bf)[0]

        a, b = ops.bottleneck_bf16(x, pack), sep()
        mism = float(((a.float() - b.float()).abs() > 0).float().mean())
        t_f, t_s = timeit(lambda: ops.bottleneck_bf16(x, pack)), timeit(sep)
        flops = 2.0 * 17 * cm * cm * H * W
        print("Cm %3d  %3dx%3d  fused %7.1f us (%5.1f TFLOP/s, %4.0f GB/s in+out)  three launches %7.1f us   mismatch %.2e"
              % (cm, H, W, t_f, flops / t_f * 1e-6, 4.0 * c * H * W / t_f * 1e-3, t_s, mism), flush=True)


def main_proj():
    """The first (projection) bottleneck of res2 / res3 / res4: one launch against the four it replaces."""
    bf = torch.bfloat16
    for cm, cin, st, H, W in [(64, 64, 1, 256, 512), (128, 256, 2, 256, 512), (256, 512, 2, 128, 256)]:
        c = 4 * cm
        g = torch.Generator().manual_seed(cm)
        mk = lambda *sh: (torch.randn(*sh, generator=g) / (sh[1] * sh[2] * sh[3]) ** 0.5).cuda()
        w1, w2, w3, wd = mk(cm, cin, 1, 1), mk(cm, cm, 3, 3), mk(c, cm, 1, 1), mk(c, cin, 1, 1)
        b1, b2, b3, bd = torch.zeros(cm).cuda(), torch.zeros(cm).cuda(), torch.zeros(c).cuda(), torch.zeros(c).cuda()
        x = torch.randn(1, cin, H, W, device='cuda').to(bf).contiguous(memory_format=torch.channels_last)
        pack = ops.pack_bottleneck_proj_bf16(w1, w2, w3, wd, b1, b2, b3, bd)
        p1, p2, p3, pd = (ops.pack_conv_weight_bf16(w, split=False) for w in (w1, w2, w3, wd))

        def sep():
            t1 = ops.conv2d_nhwc_bf16_multi([x], p1[0], None, p1[2], b1, cm, 1, st, 0, relu=True, out_dtype=bf)[0]
            t2 = ops.conv2d_nhwc_bf16_multi([t1], p2[0], None, p2[2], b2, cm, 3, 1, 1, relu=True, out_dtype=bf)[0]
            sc = ops.conv2d_nhwc_bf16_multi([x], pd[0], None, pd[2], bd, c, 1, st, 0, relu=False, out_dtype=bf)[0]
            return ops.conv2d_nhwc_bf16_multi([t2], p3[0], None, p3[2], b3, c, 1, 1, 0, relu=True, residuals=[sc], out_dtype=bf)[0]

        t_f, t_s = timeit(lambda: ops.bottleneck_proj_bf16(x, pack, st)), timeit(sep)
        Ho, Wo = (H - 1) // st + 1, (W - 1) // st + 1
        print("projection block Cm %3d  Cin %3d /%d  %3dx%3d  fused %7.1f us   four launches %7.1f us" % (cm, cin, st, Ho, Wo, t_f, t_s), flush=True)


if __name__ == '__main__':
    main()
    main_proj()
