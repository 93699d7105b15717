"""Times the dense kernels at the model's shapes (batch 8): GB/s over the fp32 weights."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gansynth_amd import kernels
K = kernels.get()
def t(fn, n=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for (b, i, o) in [(8, 512, 8192), (8, 8192, 256), (8, 256, 61), (16, 8192, 256)]:
    x = torch.randn(b, i, device="cuda").to(torch.bfloat16)
    gy = torch.randn(b, o, device="cuda").to(torch.bfloat16)
    w = torch.randn(i, o, device="cuda")
    gw = torch.zeros(i, o, device="cuda")
    mb = i * o * 4 / 1e6
    f = t(lambda: K.dense_fwd(x, w, 0.1)); d = t(lambda: K.dense_bwd_data(gy, w, 0.1)); g = t(lambda: K.dense_bwd_weight(x, gy, 0.1, out=gw))
    print("b %2d %5d -> %5d (%.1f MB of weights): fwd %5.1f us (%4.0f GB/s)  bwd_data %5.1f us (%4.0f GB/s)  bwd_weight %5.1f us (%4.0f GB/s r+w)" %
          (b, i, o, mb, f, mb / f * 1e3, d, mb / d * 1e3, g, 2 * mb / g * 1e3))
