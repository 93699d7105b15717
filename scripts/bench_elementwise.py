"""Achieved HBM rate of the elementwise / reduction kernels at the activation sizes of the fully grown networks
(batch 8, bf16 unless --f32): each op is launched back to back between two events; bytes = tensors read + written."""
import sys
import torch
from gansynth_amd import kernels

K = kernels.get()
dev = torch.device("cuda:0")
dt = torch.float32 if "--f32" in sys.argv else torch.bfloat16
LEVELS = [(8 * 128 * 1024, 32), (8 * 64 * 512, 64), (8 * 32 * 256, 128), (8 * 16 * 128, 256)]
ACT_LRELU = 1


def timed(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def line(name, us, tensors, p, c):
    nbytes = tensors * p * c * (4 if dt == torch.float32 else 2)
    print(f"  {name:44s} {us:7.1f} us  {nbytes / us / 1e6:6.2f} TB/s  ({tensors} tensors)")


for p, c in LEVELS:
    print(f"p = {p}, c = {c}  ({p * c * 2 / 2**20:.0f} MiB per bf16 tensor)")
    x = torch.randn(p, c, device=dev).to(dt)
    g = torch.randn(p, c, device=dev).to(dt)
    gg = torch.randn(p, c, device=dev).to(dt)
    mk = lambda t: t.view(8, -1, 1, c).permute(0, 3, 1, 2)   # logical NCHW over channels-last memory
    x4, g, gg = mk(x), mk(g), mk(gg)
    line("torch add_ (reference stream)", timed(lambda: g.add_(1.0)), 2, p, c)
    line("pixel_norm_fwd", timed(lambda: K.pixel_norm_fwd(x4, 1e-8)), 2, p, c)
    line("pixel_norm_bwd", timed(lambda: K.pixel_norm_bwd(g, x4, 1e-8)), 3, p, c)
    line("pixel_norm_bwd act+pre", timed(lambda: K.pixel_norm_bwd(g, x4, 1e-8, act=ACT_LRELU, pre_act=0)), 3, p, c)
    line("pixel_norm_bwd act + addend", timed(lambda: K.pixel_norm_bwd(g, x4, 1e-8, act=ACT_LRELU, addend=gg)), 4, p, c)
    line("pixel_norm_bwd_bwd", timed(lambda: K.pixel_norm_bwd_bwd(gg, g, x4, 1e-8)), 4, p, c)
    line("pixel_norm_bwd_bwd with_g", timed(lambda: K.pixel_norm_bwd_bwd(gg, g, x4, 1e-8, pre_act=ACT_LRELU, with_g=True)), 5, p, c)
    line("act_bwd", timed(lambda: K.act_bwd(g, x4, ACT_LRELU)), 3, p, c)
    line("act_bwd_bias", timed(lambda: K.act_bwd_bias(g, x4, ACT_LRELU)), 3, p, c)
    line("channel_sum", timed(lambda: K.channel_sum(x4)), 1, p, c)
    line("bias_act_fwd", timed(lambda: K.bias_act_fwd(x4, None, ACT_LRELU)), 2, p, c)
    line("axpby", timed(lambda: K.axpby(x4, g, 0.5, 0.5)), 3, p, c)
    line("sumsq_rows (8 rows)", timed(lambda: K.sumsq_rows(x.view(8, -1))), 1, p, c)
