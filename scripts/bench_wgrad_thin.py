"""Times the weight gradients of the two 32-channel top-level layers (conv_wgrad_bf16_kernel: HBM-bound, 144 flop/byte): 16 / 24 images."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gansynth_amd import kernels
K = kernels.get()
CL = torch.channels_last
for n in (16, 24):
    for (ci, co, st) in ((32, 32, 1), (32, 64, 2)):
        x = torch.randn(n, ci, 128, 1024, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
        gy = torch.randn(n, co, 128 // st, 1024 // st, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
        gw = torch.zeros(3, 3, ci, co, device="cuda"); gb = torch.zeros(co, device="cuda")
        f = lambda: K.conv2d_bwd_weight(x, gy, 3, st, 0.1, out=gw, bias_out=gb)
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e9
        for _ in range(5):
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20)
        mb = (x.numel() + gy.numel()) * 2 / 1e6
        print("%d images %d->%d stride %d: %.1f us (launch + fold), %.0f GB/s over x + gy" % (n, ci, co, st, best * 1e3, mb / best))
