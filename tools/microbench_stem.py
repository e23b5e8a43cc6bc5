"""Time the backbone stem at 1024x2048: the one-launch kernels (csrc/stem_pool.hip fp32, csrc/stem_pool_bf16.hip) against the stem
convolution kernel + library max-pool. GPU only."""
import sys
import torch
import torch.nn.functional as F

sys.path.insert(0, '.')
from upsnet_amd import ops  # noqa: E402


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
    for H, W in ((1024, 2048), (800, 1344)):
        x4 = ops.image_to_nhwc4(torch.randn(1, 3, H, W, device='cuda') * 50)
        w = torch.randn(64, 3, 7, 7, device='cuda') * 0.02
        b = torch.randn(64, device='cuda')
        wp, ldw = ops.pack_stem_weight(w)
        w32, w16 = ops.pack_stem_pool_weight_f32(w), ops.pack_stem_pool_weight_bf16(w)
        t_conv = timeit(lambda: ops.conv2d_stem(x4, wp, ldw, b, 64, 7, 7, 2, 3, relu=True))
        y = ops.conv2d_stem(x4, wp, ldw, b, 64, 7, 7, 2, 3, relu=True)
        t_pool = timeit(lambda: F.max_pool2d(y, 3, stride=2, padding=1))
        t_f32 = timeit(lambda: ops.stem_pool_f32(x4, w32, b))
        t_b16 = timeit(lambda: ops.stem_pool_bf16(x4, w16, b))
        print("%dx%d  stem kernel %.1f us + library max-pool %.1f us   fused fp32 %.1f us   fused bf16 %.1f us" % (H, W, t_conv, t_pool, t_f32, t_b16), flush=True)


if __name__ == '__main__':
    main()
