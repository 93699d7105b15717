"""Per-launch time of one conv layer launched 100 times back to back: eager launches vs one captured hipGraph (what the step uses).
The difference to the in-kernel time of scripts/probe/igemm_trace is the dispatch gap a graph node still pays."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gansynth_amd import kernels

K = kernels.get()
dt = torch.bfloat16
for name, ci, co, h, w in (("s1 128->128 @32x256", 128, 128, 32, 256), ("s1 32->32 @128x1024", 32, 32, 128, 1024), ("s1 256->256 @4x32", 256, 256, 4, 32)):
    x = torch.randn(8, ci, h, w, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(3, 3, ci, co, device="cuda")
    K.register_param_buffer(wt)
    f = lambda: K.conv2d_fwd(x, wt, 3, 1, 0.1)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(100):
            f()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    tg = (time.perf_counter() - t0) / 1000 * 1e6
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        f()
    e1.record()
    torch.cuda.synchronize()
    te = e0.elapsed_time(e1) * 1e3 / 100
    print(f"{name:24s} graph node {tg:6.2f} us   eager back-to-back {te:6.2f} us")
