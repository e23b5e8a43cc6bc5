"""Micro-benchmark of the fp32 MFMA conv / fused DCN kernels on the C1 shapes (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from upsnet_amd import ops

torch.manual_seed(0)
from upsnet_amd._lib import lib
lib().upsnet_conv_tuning(int(os.environ.get('WINO_TILES', '0')), int(os.environ.get('CONV_TILE', '0')))
print('winograd tiles', os.environ.get('WINO_TILES', '0'), 'tile', os.environ.get('CONV_TILE', '0'), flush=True)
reps = int(os.environ.get('REPS', '10'))
def bench(name, fn, flops, bytes_):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("%-42s %8.3f ms  %7.1f TFLOP/s  %7.1f GB/s" % (name, ms, flops / ms / 1e9, bytes_ / ms / 1e6), flush=True)

cases = [  # name, Cin, Cout, H, W, k, stride
    ("fpn_p2 3x3 256->256 @256x512", 256, 256, 256, 512, 3, 1),
    ("res2 conv1 1x1 256->64 @256x512", 256, 64, 256, 512, 1, 1),
    ("res2 conv2 3x3 64->64 @256x512", 64, 64, 256, 512, 3, 1),
    ("res2 conv3 1x1 64->256 @256x512", 64, 256, 256, 512, 1, 1),
    ("res3 conv2 3x3 128->128 @128x256", 128, 128, 128, 256, 3, 1),
    ("res4 conv2 3x3 256->256 @64x128", 256, 256, 64, 128, 3, 1),
    ("res4 conv3 1x1 256->1024 @64x128", 256, 1024, 64, 128, 1, 1),
    ("res5 conv2 3x3 512->512 @32x64", 512, 512, 32, 64, 3, 1),
    ("fpn lat 1x1 2048->256 @32x64", 2048, 256, 32, 64, 1, 1),
    ("fpn lat 1x1 256->256 @256x512", 256, 256, 256, 512, 1, 1),
    ("res3 conv3 1x1 128->512 @128x256", 128, 512, 128, 256, 1, 1),
    ("res3 conv1 1x1 512->128 @128x256", 512, 128, 128, 256, 1, 1),
    ("res4 conv1 1x1 1024->256 @64x128", 1024, 256, 64, 128, 1, 1),
    ("res3 down 1x1/2 256->512 @256x512", 256, 512, 256, 512, 1, 2),
    ("res5 conv3 1x1 512->2048 @32x64", 512, 2048, 32, 64, 1, 1),
    ("res5 conv1 1x1 2048->512 @32x64", 2048, 512, 32, 64, 1, 1),
]
only = os.environ.get('ONLY')
for name, cin, cout, H, W, k, st in cases:
    if only and only not in name: continue
    x = torch.randn(1, cin, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
    w = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    b = torch.randn(cout, device='cuda')
    wp, ldw = ops.pack_conv_weight(w)
    Ho, Wo = H // st, W // st
    fl = 2.0 * cout * cin * k * k * Ho * Wo
    by = 4.0 * (cin * H * W + cout * Ho * Wo + cout * cin * k * k)
    bench("hip  " + name, lambda: ops.conv2d_nhwc(x, wp, ldw, b, cout, k, st, k // 2, relu=True), fl, by)
    if k == 1 and cin % 32 == 0 and cout >= 32:
        wf = ops.pack_conv1x1_weight(w)
        for t in (64, 128):
            lib().upsnet_conv1x1_tuning(t)
            bench("gemm%-3d %s" % (t, name), lambda: ops.conv1x1_frag(x, wf, b, cout, st, relu=True), fl, by)
        lib().upsnet_conv1x1_tuning(0)
    if not os.environ.get('NOTORCH'):
        wt = w.contiguous(memory_format=torch.channels_last)
        bench("torch " + name, lambda: F.relu(F.conv2d(x, wt, b, stride=st, padding=k // 2)), fl, by)

if not only or 'dcn' in only:
    sizes = [(256, 512), (128, 256), (64, 128), (32, 64)]
    for cin, cout in ((256, 128), (128, 128)):
        xs = [torch.randn(1, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last) for h, w in sizes]
        offs = [(torch.randn(1, 18, h, w, device='cuda') * 2).contiguous(memory_format=torch.channels_last) for h, w in sizes]
        wgt = torch.randn(cout, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
        hw = sum(h * w for h, w in sizes)
        for std in (2.0, 0.3):   # offset spread in pixels (the synthetic benchmark weights give ~0.3-3 px)
            offs = [(torch.randn(1, 18, h, w, device='cuda') * std).contiguous(memory_format=torch.channels_last) for h, w in sizes]
            for kind, variants in (('frag_bf16', (None,)), ('frag', (5, 6))):
                wp = ops.pack_dcn_weight(wgt, kind)
                for v in variants:
                    if v is not None:
                        lib().upsnet_dcn_tuning(v)
                    bench("dcn %s%s %d->%d 4 levels, offsets N(0,%.1f)" % (kind, '' if v is None else ' v%d' % v, cin, cout, std),
                          lambda: ops.deform_conv_fused(xs, offs, wp, None, cin, cout, (3, 3), (1, 1), (1, 1), (1, 1), relu=True),
                          2.0 * cout * cin * 9 * hw, 4.0 * hw * (cin + 18 + cout))
            lib().upsnet_dcn_tuning(0)

if not only or 'pair' in only:   # block boundary of res2: conv3 + shortcut + ReLU, then the next conv1 + ReLU -- two launches vs one
    for H, W, c1 in ((256, 512, 256),):
        x = torch.randn(1, 64, H, W, device='cuda').relu().contiguous(memory_format=torch.channels_last)
        sc = torch.randn(1, c1, H, W, device='cuda').contiguous(memory_format=torch.channels_last)
        p3 = ops.pack_conv1x1_weight(torch.randn(c1, 64, 1, 1, device='cuda') / 8)
        p1 = ops.pack_conv1x1_weight(torch.randn(64, c1, 1, 1, device='cuda') / c1 ** 0.5)
        b3, b1 = torch.randn(c1, device='cuda'), torch.randn(64, device='cuda')
        fl = 2.0 * H * W * (64 * c1 + c1 * 64)
        def two():
            o1 = ops.conv1x1_frag(x, p3, b3, c1, 1, relu=True, residual=sc)
            return ops.conv1x1_frag(o1, p1, b1, 64, 1, relu=True)
        bench("res2 conv3 + next conv1, two launches", two, fl, 4.0 * H * W * (64 + 3 * c1 + 64))
        bench("res2 conv3 + next conv1, pair kernel", lambda: ops.conv1x1_pair(x, sc, p3, b3, c1, p1, b1, 64), fl, 4.0 * H * W * (64 + 2 * c1 + 64))
