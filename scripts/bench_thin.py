"""Micro-benchmark of the colour blocks' 1x1 convs at the top level (batch 8, 128x1024): features <- colour (2 -> 32, bias + leaky relu)
and colour <- features (32 -> 2, bias + tanh).  usage: python scripts/bench_thin.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gansynth_amd import _lib, kernels

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
K = kernels.get()
CL = torch.channels_last
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for ci, co, act in [(2, 32, _lib.ACT_LRELU), (32, 2, _lib.ACT_TANH)]:
    x = torch.randn(8, ci, 128, 1024, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=CL)
    w = torch.randn(1, 1, ci, co, device="cuda", generator=g)
    b = torch.randn(co, device="cuda", generator=g)
    us = timeit(lambda: K.conv2d_fwd_bias_act(x, w, b, 1, 1, 0.5, act))
    mb = (x.numel() + 8 * co * 128 * 1024) * 2 / 1e6
    print("1x1 %2d -> %2d @128x1024 x8: %6.1f us  (%5.1f MB, %.2f TB/s)" % (ci, co, us, mb, mb / us / 1e6 * 1e6 / 1e6))
