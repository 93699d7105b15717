"""Micro-benchmark of the conv entry points on the layer shapes of the fully grown PGGAN (batch 8).
usage: python scripts/bench_conv.py [bf16|f32] [filter-substring] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gansynth_amd import kernels

dt = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
flt = sys.argv[2] if len(sys.argv) > 2 else ""
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 20
K = kernels.get()
B = 8
# name, kind, ci, co, h, w (input side dims)
LAYERS = [
    ("s1_32x32@128x1024", "s1", 32, 32, 128, 1024),
    ("s1_64x64@64x512", "s1", 64, 64, 64, 512),
    ("s1_128x128@32x256", "s1", 128, 128, 32, 256),
    ("s1_256x256@16x128", "s1", 256, 256, 16, 128),
    ("s1_256x256@8x64", "s1", 256, 256, 8, 64),
    ("s2_32to64@128x1024", "s2", 32, 64, 128, 1024),
    ("s2_64to128@64x512", "s2", 64, 128, 64, 512),
    ("s2_128to256@32x256", "s2", 128, 256, 32, 256),
    ("s2_256to256@16x128", "s2", 256, 256, 16, 128),
    ("t2_64to32@64x512", "t2", 64, 32, 64, 512),
    ("t2_128to64@32x256", "t2", 128, 64, 32, 256),
    ("t2_256to128@16x128", "t2", 256, 128, 16, 128),
    ("t2_256to256@8x64", "t2", 256, 256, 8, 64),
]


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3  # us


print(f"dtype {dt}, batch {B}; columns: us (TFLOP/s, GB/s of in+out)")
for name, kind, ci, co, h, w in LAYERS:
    if flt and flt not in name:
        continue
    x = torch.randn(B, ci, h, w, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(3, 3, ci, co, device="cuda")
    if kind == "t2":
        oh, ow = 2 * h, 2 * w
        fwd = lambda: K.conv2d_transpose_fwd(x, wt, 0.1)
    else:
        st = 2 if kind == "s2" else 1
        oh, ow = h // st, w // st
        fwd = lambda: K.conv2d_fwd(x, wt, 3, st, 0.1)
    y = fwd()
    gy = torch.randn_like(y)
    if kind == "t2":
        bwd_d = lambda: K.conv2d_transpose_bwd_data(gy, wt, 0.1)
        bwd_w = lambda: K.conv2d_transpose_bwd_weight(x, gy, 0.1)
        flops = 2.0 * 9 * B * h * w * ci * co
    else:
        bwd_d = lambda: K.conv2d_bwd_data(gy, wt, x.shape, 3, st, 0.1)
        bwd_w = lambda: K.conv2d_bwd_weight(x, gy, 3, st, 0.1)
        flops = 2.0 * 9 * B * oh * ow * ci * co
    nbytes = (x.numel() + y.numel()) * x.element_size()
    out = []
    for lab, fn in (("fwd", fwd), ("bwd_data", bwd_d), ("bwd_weight", bwd_w)):
        us = timeit(fn)
        out.append(f"{lab} {us:7.1f}us ({flops / us / 1e6:6.1f} TF, {nbytes / us / 1e3:6.0f} GB/s)")
    print(f"{name:22s} " + " | ".join(out), flush=True)
