import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from upsnet_amd.config.config import update_config_dict, CITYSCAPES_R50
update_config_dict(CITYSCAPES_R50)
from upsnet_amd.synthetic import build_model, make_image
model = build_model()
imgs = [make_image(1024, 2048, seed=j, device='cuda') for j in range(2)]
with torch.no_grad():
    for _ in range(8):
        model(imgs[0])
    torch.cuda.synchronize()
    slots = next(iter(model._graphs.values()))['slots']
    assert all('graph' in s for s in slots), [list(s.keys()) for s in slots]
    g0, g1 = slots[0]['graph'], slots[1]['graph']
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    def serial(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n):
            (g0 if i % 2 == 0 else g1).replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    def conc(n):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(n // 2):
            with torch.cuda.stream(s1): g0.replay()
            with torch.cuda.stream(s2): g1.replay()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
    for _ in range(2):
        print("serial  %.3f ms/img" % serial(40), flush=True)
        print("2-stream %.3f ms/img" % conc(40), flush=True)
