"""Fused data gradient + previous block's pixel-norm backward (gs_conv2d[_transpose_s2]_bwd_data_pnbwd) against the two separate launches,
on the three full-size shapes that have the epilogue form.  Burst timing (50 back-to-back calls between one event pair)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gansynth_amd import kernels
K = kernels.get()
CL = torch.channels_last
dt = torch.bfloat16


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for kind, n, ci, co, h, w in [("conv", 8, 32, 32, 128, 1024), ("conv", 8, 64, 64, 64, 512), ("convT", 8, 64, 32, 64, 512), ("conv1", 8, 32, 2, 128, 1024)]:
    z = torch.randn(n, ci, h, w, device="cuda").to(dt).contiguous(memory_format=CL)
    add = torch.randn(n, ci, h, w, device="cuda").to(dt).contiguous(memory_format=CL)
    ks = 1 if kind == "conv1" else 3
    wt = torch.randn(ks, ks, ci, co, device="cuda")
    K.register_param_buffer(wt)
    oh, ow = (2 * h, 2 * w) if kind == "convT" else (h, w)
    gy = torch.randn(n, co, oh, ow, device="cuda").to(dt).contiguous(memory_format=CL)
    if kind != "convT":
        sep = lambda: K.pixel_norm_bwd(K.conv2d_bwd_data(gy, wt, (n, ci, h, w), ks, 1, 0.05), z, 1e-8, act=1, addend=add)
        fus = lambda: K.conv2d_bwd_data_pnbwd(gy, wt, (n, ci, h, w), ks, 1, 0.05, z, 1e-8, 1, addend=add)
        conv = lambda: K.conv2d_bwd_data(gy, wt, (n, ci, h, w), ks, 1, 0.05)
    else:
        sep = lambda: K.pixel_norm_bwd(K.conv2d_transpose_bwd_data(gy, wt, 0.05), z, 1e-8, act=1, addend=add)
        fus = lambda: K.conv2d_transpose_bwd_data_pnbwd(gy, wt, 0.05, z, 1e-8, 1, addend=add)
        conv = lambda: K.conv2d_transpose_bwd_data(gy, wt, 0.05)
    print("%s %d->%d @ %dx%d: conv alone %.1f us, conv + norm backward %.1f us, fused %.1f us" % (kind, co, ci, h, w, timed(conv), timed(sep), timed(fus)))
