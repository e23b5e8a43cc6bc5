"""Per-launch table of the convolution kernels over one eager forward of the bench workload (development aid):
time, algorithmic TFLOP/s and algorithmic GB/s of every launch, sorted by time, from the ops.PROFILE events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('UPSNET_GRAPH', '0')
os.environ.setdefault('UPSNET_OVERLAP', '0')
import torch
from upsnet_amd import ops
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50, COCO_R101_DCN
# usage: layer_table.py [c3 | c4]   (default: UPSNet-50 1024x2048; c3: UPSNet-101-DCN 800x1333; c4: UPSNet-101-DCN 1024x2048)
which = sys.argv[1] if len(sys.argv) > 1 else 'c1'
update_config_dict(CITYSCAPES_R50 if which == 'c1' else COCO_R101_DCN)
from upsnet_amd.synthetic import build_model, make_image

data = make_image(800, 1333, seed=0, device='cuda') if which == 'c3' else make_image(1024, 2048, seed=0, device='cuda')
model = build_model()
with torch.no_grad():
    for _ in range(3):
        model(data)
    torch.cuda.synchronize()
    acc = {}
    for it in range(5):
        ops.PROFILE['events'] = []
        ops.PROFILE['enabled'] = True
        model(data)
        ops.PROFILE['enabled'] = False
        torch.cuda.synchronize()
        for i, e in enumerate(ops.PROFILE['events']):
            key = (i, e[0], e[5] if len(e) > 5 else e[0])
            acc.setdefault(key, [0.0, e[3], e[4]])[0] += e[1].elapsed_time(e[2]) / 5
rows = sorted(acc.items(), key=lambda kv: -kv[1][0])
tot = sum(v[0] for v in acc.values())
print("%d launches, %.3f ms total" % (len(rows), tot))
for (i, kind, desc), (ms, fl, by) in rows:
    print("%7.1f us %5.1f%% | %6.1f TF %7.0f GB/s | #%-3d %s" % (ms * 1e3, 100 * ms / tot, fl / ms / 1e9, by / ms / 1e6, i, desc))
