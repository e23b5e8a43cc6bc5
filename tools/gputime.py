"""GPU-time measurement for short kernels: the launches are captured into one HIP graph and the replay is timed, so the host's per-call
cost (Python + ctypes + torch.empty: 20-40 us) is not in the number. gpu_time(fn, n) -> microseconds per call."""
import torch


def gpu_time(fn, n=20, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.graph(g, stream=s):
        for _ in range(n):
            fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1000.0 / n)
    return sorted(ts)[len(ts) // 2]
