"""Second-order form: forward conv on a cotangent + both gradients of the block's norm-backward node (gs_conv2d[_transpose_s2]_fwd_pnbwdbwd)
against the two separate launches (conv, gs_pixel_norm_bwd_bwd_fused), on the four full-size shapes with the epilogue form.  Burst timing."""
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


for kind, n, ci, co, h, w in [("conv", 8, 32, 32, 128, 1024), ("convT", 8, 64, 32, 64, 512), ("conv", 8, 64, 64, 64, 512), ("convT", 8, 128, 64, 32, 256)]:
    x = torch.randn(n, ci, h, w, device="cuda").to(dt).contiguous(memory_format=CL)
    wt = torch.randn(3, 3, ci, co, device="cuda")
    K.register_param_buffer(wt)
    oh, ow = (2 * h, 2 * w) if kind == "convT" else (h, w)
    z = torch.randn(n, co, oh, ow, device="cuda").to(dt).contiguous(memory_format=CL)
    g = torch.randn(n, co, oh, ow, device="cuda").to(dt).contiguous(memory_format=CL)
    if kind == "conv":
        conv = lambda: K.conv2d_fwd(x, wt, 3, 1, 0.05)
        sep = lambda: K.pixel_norm_bwd_bwd(K.conv2d_fwd(x, wt, 3, 1, 0.05), g, z, 1e-8, pre_act=1, with_g=True)
        fus = lambda: K.conv2d_fwd_pnbwdbwd(x, wt, 3, 1, 0.05, g, z, 1e-8, 1)
    else:
        conv = lambda: K.conv2d_transpose_fwd(x, wt, 0.05)
        sep = lambda: K.pixel_norm_bwd_bwd(K.conv2d_transpose_fwd(x, wt, 0.05), g, z, 1e-8, pre_act=1, with_g=True)
        fus = lambda: K.conv2d_transpose_fwd_pnbwdbwd(x, wt, 0.05, g, z, 1e-8, 1)
    print("%s %d->%d @ %dx%d: conv alone %.1f us, conv + second-order norm %.1f us, fused %.1f us (fused form: %s)" % (
        kind, ci, co, h, w, timed(conv), timed(sep), timed(fus), K.fwd_pnbwdbwd_is_fused((n, ci, h, w), co, 3, 2 if kind == "convT" else 1, kind == "convT", dt)))
