"""Masked data-gradient / forward convs with the activation tensor as the mask against its sign words (GS_MASK_BITS), launch-to-launch."""
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


for n, ci, co, h, w, s in [(8, 32, 32, 128, 1024, 1), (8, 32, 64, 128, 1024, 2), (8, 64, 64, 64, 512, 1), (8, 64, 128, 64, 512, 2), (8, 128, 128, 32, 256, 1), (8, 256, 256, 16, 128, 1)]:
    x = torch.randn(n, ci, h, w, device="cuda").to(dt).contiguous(memory_format=CL)      # the masked conv's forward input = the mask
    bits = torch.randint(-32768, 32767, (n, h, w, ci // 16), device="cuda", dtype=torch.int16)
    wt = torch.randn(3, 3, ci, co, device="cuda")
    K.register_param_buffer(wt)
    gy = torch.randn(n, co, h // s, w // s, device="cuda").to(dt).contiguous(memory_format=CL)
    plain = lambda: K.conv2d_bwd_data(gy, wt, (n, ci, h, w), 3, s, 0.05)
    full = lambda: K.conv2d_bwd_data(gy, wt, (n, ci, h, w), 3, s, 0.05, mask=x, mask_act=1)
    bit = lambda: K.conv2d_bwd_data(gy, wt, (n, ci, h, w), 3, s, 0.05, mask_act=1, mask_bits=bits)
    fwd = lambda: K.conv2d_fwd_bias_act(x, wt, None, 3, s, 0.05, 1)
    fwdb = lambda: K.conv2d_fwd_bias_act(x, wt, None, 3, s, 0.05, 1, want_bits=True)
    print("data gradient %d <- %d @ %dx%d / %d: plain %.1f us, tensor mask %.1f us, sign words %.1f us; forward %.1f us, forward + sign words %.1f us"
          % (ci, co, h, w, s, timed(plain), timed(full), timed(bit), timed(fwd), timed(fwdb)))
