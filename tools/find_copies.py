"""List the small copy / fill / elementwise launches of one forward with their Python call sites (development aid)."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
update_config_dict(CITYSCAPES_R50)
from upsnet_amd.synthetic import build_model, make_image
model = build_model()
data = make_image(1024, 2048, seed=0, device='cuda')
with torch.no_grad():
    for _ in range(4): model(data)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        model(data)
        torch.cuda.synchronize()
pat = sys.argv[1:] or ['aten::copy_', 'aten::fill_', 'aten::zero_', 'aten::clone', 'aten::contiguous', 'aten::to', 'aten::_to_copy', 'aten::zeros', 'aten::cat', 'aten::clamp', 'aten::relu', 'aten::add', 'aten::mul', 'aten::sub', 'aten::div', 'aten::index', 'aten::sigmoid']
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in pat:
        site = next((s for s in ev.stack if 'upsnet_amd' in s and 'find_copies' not in s), None)
        if site is None: continue
        cnt[(ev.name, site.split('upsnet_amd/')[-1][:80])] += 1
for (name, site), c in sorted(cnt.items(), key=lambda kv: -kv[1])[:70]:
    print("%3d %-18s %s" % (c, name, site))
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
