"""Time the 3x3 256 -> 256 layers of the bf16 mode at the shapes of the 1024x2048 workload: the general haloed-patch kernel against
csrc/conv3x3_wreg_bf16.hip (weights from L2 into the MFMA) at both tile heights. GPU only."""
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
    cases = {
        'FPN out P2-P5': [(1, 256, 512), (1, 128, 256), (1, 64, 128), (1, 32, 64)],
        'FPN out P2': [(1, 256, 512)],
        'RPN P2-P6': [(1, 256, 512), (1, 128, 256), (1, 64, 128), (1, 32, 64), (1, 16, 32)],
        'mask head 100 ROIs': [(100, 14, 14)],
        'mask head 40 ROIs': [(40, 14, 14)],
        'res4 conv2': [(1, 64, 128)],
        'res5 conv2 (512)': [(1, 32, 64)],
    }
    for name, shapes in cases.items():
        C = 512 if '512' in name else 256
        w = (torch.randn(C, C, 3, 3) / (9 * C) ** 0.5).cuda()
        b = torch.randn(C).cuda()
        hi, _, ldw = ops.pack_conv_weight_bf16(w, split=False)
        for in16 in (False, True):
            xs = [torch.randn(n, C, h, w_, device='cuda').contiguous(memory_format=torch.channels_last) for n, h, w_ in shapes]
            if in16:
                xs = [x.bfloat16() for x in xs]
            px = sum(n * h * w_ for n, h, w_ in shapes)
            row = []
            for en, th in ((0, 0), (1, 1), (1, 2), (1, 8), (1, 16), (1, 0)):
                lib().upsnet_conv_bf16_tuning(en, th)
                t = timeit(lambda: ops.conv2d_nhwc_bf16_multi(xs, hi, None, ldw, b, C, 3, 1, 1, relu=True))
                row.append(t)
            lib().upsnet_conv_bf16_tuning(1, 0)
            fl = 2.0 * 9 * C * C * px
            print("%-20s %s in   halo %7.1f us   wreg2/128 %7.1f   wreg2 %7.1f   wreg8 %7.1f   wreg16 %7.1f   auto %7.1f us (%5.0f TFLOP/s)"
                  % (name, 'bf16' if in16 else 'fp32', row[0], row[1], row[2], row[3], row[4], row[5], fl / row[5] * 1e-6), flush=True)


if __name__ == '__main__':
    main()
