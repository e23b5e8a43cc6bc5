"""A/B the K walk order of the fp32 MFMA conv (slab-outer vs tap-outer) over the k>1 conv shapes + the fused DCN (development aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
update_config_dict(CITYSCAPES_R50)
from upsnet_amd.synthetic import build_model, make_image
from upsnet_amd import ops
from upsnet_amd._lib import lib

shapes = {}
orig = ops.conv2d_nhwc_multi
def rec(xs, wpack, ldw, bias, cout, ksize, stride, pad, relu=False, residuals=None):
    key = (tuple((x.shape[0], x.shape[2], x.shape[3]) for x in xs), xs[0].shape[1], cout, ksize, stride, residuals is not None)
    shapes[key] = shapes.get(key, 0) + 1
    return orig(xs, wpack, ldw, bias, cout, ksize, stride, pad, relu, residuals)
ops.conv2d_nhwc_multi = rec
model = build_model(cls_gain=0.3)
data = make_image(1024, 2048, seed=0, device='cuda')
with torch.no_grad():
    model(data)
ops.conv2d_nhwc_multi = orig

def timeit(fn, n=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1000

tot = [0.0, 0.0]
for key, cnt in sorted(shapes.items(), key=lambda kv: -kv[1]):
    segs, cin, cout, k, st, has_res = key
    if k == 1: continue
    xs = [torch.randn(n, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    wgt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    b = torch.randn(cout, device='cuda')
    wp, ldw = ops.pack_conv_weight(wgt)
    us = []
    for pipe in (1, 3):
        lib().upsnet_conv_tuning(pipe, 0)
        us.append(timeit(lambda: ops.conv2d_nhwc_multi(xs, wp, ldw, b, cout, k, st, k // 2, True, None)))
    tot[0] += us[0] * cnt; tot[1] += us[1] * cnt
    print("x%-2d %s %4d->%-4d k%d s%d | slab-outer %.0f us, tap-outer %.0f us" % (cnt, segs, cin, cout, k, st, us[0], us[1]), flush=True)
print("dense k>1 total per image: slab-outer %.2f ms, tap-outer %.2f ms" % (tot[0] / 1000, tot[1] / 1000))
# fused DCN, FCN head layer 0 (256->128) and layer 1 (128->128) over P2..P5
for cin in (256, 128):
    xs = [torch.randn(1, cin, 256 >> l, 512 >> l, device='cuda').contiguous(memory_format=torch.channels_last) for l in range(4)]
    offs = [(2.0 * torch.randn(1, 18, 256 >> l, 512 >> l, device='cuda')).contiguous(memory_format=torch.channels_last) for l in range(4)]
    wgt = torch.randn(128, cin, 3, 3, device='cuda') / (cin * 9) ** 0.5
    pk = ops.pack_conv_weight(wgt)
    for tile in (0, 5, 6):
        us = []
        for pipe in (1, 3):
            lib().upsnet_conv_tuning(pipe, tile)
            us.append(timeit(lambda: ops.deform_conv_fused(xs, offs, pk, None, cin, 128, (3, 3), (1, 1), (1, 1), (1, 1), None, True)))
        print("DCN %d->128 tile %d | slab-outer %.0f us, tap-outer %.0f us" % (cin, tile, us[0], us[1]), flush=True)
lib().upsnet_conv_tuning(-1, 0)
