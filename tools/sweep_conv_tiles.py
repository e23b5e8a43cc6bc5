"""Sweep tile choices of the fp32 MFMA conv over every dense-conv shape of the model (development aid)."""
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
model = build_model()
data = make_image(1024, 2048, seed=0, device='cuda')
with torch.no_grad():
    model(data)
ops.conv2d_nhwc_multi = orig
print(len(shapes), "distinct conv shapes")
tot = {}
for key, cnt in sorted(shapes.items(), key=lambda kv: -kv[1]):
    segs, cin, cout, k, st, has_res = key
    xs = [torch.randn(n, cin, h, w, device='cuda').contiguous(memory_format=torch.channels_last) for n, h, w in segs]
    wgt = torch.randn(cout, cin, k, k, device='cuda') / (cin * k * k) ** 0.5
    b = torch.randn(cout, device='cuda')
    wp, ldw = ops.pack_conv_weight(wgt)
    res = None
    if has_res:
        res = [torch.randn_like(o) for o in ops.conv2d_nhwc_multi(xs, wp, ldw, b, cout, k, st, k // 2)]
    line = []
    best = None
    for tile in (0, 1, 2, 4, 5, 6, 3):
        if tile == 1 and ldw % 128: continue
        if tile == 4 and ldw % 128: continue
        if tile in (2, 5, 6) and ldw % 64: continue
        if tile == 6 and cin % 64: continue
        lib().upsnet_conv_tuning(0, tile)
        for _ in range(2): ops.conv2d_nhwc_multi(xs, wp, ldw, b, cout, k, st, k // 2, True, res)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): ops.conv2d_nhwc_multi(xs, wp, ldw, b, cout, k, st, k // 2, True, res)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 5 * 1000
        line.append("t%d=%.0f" % (tile, us))
        tot[tile] = tot.get(tile, 0) + (us * cnt if tile == 0 else 0)
        if tile and (best is None or us < best[1]): best = (tile, us)
        if tile == 0: auto = us
    M = sum(n * ((h + 2 * (k // 2) - k) // st + 1) * ((w + 2 * (k // 2) - k) // st + 1) for n, h, w in segs)
    print("x%-2d M=%-7d %4d->%-4d k%d s%d res%d | %s | best t%d %.0f (auto %.0f, loss %.0f us x%d)" % (cnt, M, cin, cout, k, st, has_res, " ".join(line), best[0], best[1], auto, (auto - best[1]), cnt), flush=True)
    tot['best'] = tot.get('best', 0) + best[1] * cnt
    tot['auto'] = tot.get('auto', 0) + auto * cnt
lib().upsnet_conv_tuning(0, 0)
print("total per image: auto %.2f ms, best-of-tiles %.2f ms" % (tot['auto'] / 1000, tot['best'] / 1000))
