"""Micro-benchmark of the 1-bit leaky-relu masks: per-launch time of the discriminator's masked launches with the sign bits and with the values, of
the forward launch with and without the bits written, and of the packing pass.  usage: python scripts/mask_bits_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gansynth_amd import kernels

K = kernels.HipKernels()
CL = torch.channels_last
gen = torch.Generator(device="cuda").manual_seed(0)


def t(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


for (n, c, h, w, c2) in ((8, 32, 128, 1024, 32), (8, 64, 64, 512, 64), (8, 128, 32, 256, 128), (8, 256, 16, 128, 256)):
    x = torch.randn(n, c, h, w, device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
    wt = torch.randn(3, 3, c, c, device="cuda", generator=gen)
    z = K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1)
    zc = z.clone(memory_format=CL)
    gy = torch.randn(n, c2, h, w, device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
    w2 = torch.randn(3, 3, c, c2, device="cuda", generator=gen)
    gy2 = torch.randn(n, 2 * c, h // 2, w // 2, device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
    w3 = torch.randn(3, 3, c, 2 * c, device="cuda", generator=gen)
    row = {
        "fwd+bits": t(lambda: K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1)),
        "fwd": t(lambda: K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1, bits=False)),
        "bwd_data bits": t(lambda: K.conv2d_bwd_data(gy, w2, tuple(z.shape), 3, 1, 0.07, mask=z, mask_act=1)),
        "bwd_data values": t(lambda: K.conv2d_bwd_data(gy, w2, tuple(z.shape), 3, 1, 0.07, mask=zc, mask_act=1)),
        "bwd_data none": t(lambda: K.conv2d_bwd_data(gy, w2, tuple(z.shape), 3, 1, 0.07)),
        "s2 bwd_data bits": t(lambda: K.conv2d_bwd_data(gy2, w3, tuple(z.shape), 3, 2, 0.07, mask=z, mask_act=1)),
        "s2 bwd_data values": t(lambda: K.conv2d_bwd_data(gy2, w3, tuple(z.shape), 3, 2, 0.07, mask=zc, mask_act=1)),
        "fwd_mask bits": t(lambda: K.conv2d_fwd_mask(x, wt, 3, 1, 0.07, z, 1)),
        "fwd_mask values": t(lambda: K.conv2d_fwd_mask(x, wt, 3, 1, 0.07, zc, 1)),
        "pack": t(lambda: K.lib.gs_pack_act_bits(z.data_ptr(), n * h * w, c, 1, None)),
    }
    print((n, c, h, w), "  ".join("%s %.1f" % kv for kv in row.items()), flush=True)
