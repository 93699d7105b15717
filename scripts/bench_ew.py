"""Micro-benchmark of the streaming passes at the top level (batch 8, 32 channels, 128x1024, bf16). usage: bench_ew.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gansynth_amd import kernels

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
K = kernels.get()
CL = torch.channels_last
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn):
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for c, h, w in [(32, 128, 1024), (64, 64, 512)]:
    t = [torch.randn(8, c, h, w, device="cuda", generator=g).to(torch.bfloat16).contiguous(memory_format=CL) for _ in range(4)]
    gb = torch.zeros(c, device="cuda")
    mb = t[0].numel() * 2 / 1e6
    for name, fn, n in [("pixel_norm_fwd", lambda: K.pixel_norm_fwd(t[0], 1e-8), 2),
                        ("pixel_norm_bwd(act, addend)", lambda: K.pixel_norm_bwd(t[1], t[0], 1e-8, act=1, addend=t[2]), 4),
                        ("pixel_norm_bwd(act, addend, bias)", lambda: K.pixel_norm_bwd(t[1], t[0], 1e-8, act=1, addend=t[2], bias_out=gb), 4),
                        ("pixel_norm_bwd_bwd(with_g)", lambda: K.pixel_norm_bwd_bwd(t[1], t[2], t[0], 1e-8, pre_act=1, with_g=True), 5),
                        ("act_bwd", lambda: K.act_bwd(t[1], t[0], 1), 3),
                        ("act_bwd_bias", lambda: K.act_bwd_bias(t[1], t[0], 1, out=gb), 3),
                        ("channel_sum", lambda: K.channel_sum(t[1], out=gb), 1)]:
        us = timeit(fn)
        print("%-36s c=%-3d %6.1f us  %.2f TB/s" % (name, c, us, n * mb / us))
