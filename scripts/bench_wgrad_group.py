"""Times one grouped weight-gradient flush (gs_conv_wgrad_jobs) over the stride-1 layers of the discriminator run: 16 images per layer.
usage: bench_wgrad_group.py [images] [stride]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gansynth_amd import kernels
K = kernels.get()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
st = int(sys.argv[2]) if len(sys.argv) > 2 else 1
CL = torch.channels_last
layers = [(256, 256, 4, 32), (256, 256, 8, 64), (256, 256, 16, 128), (128, 128, 32, 256), (64, 64, 64, 512)]
if st == 2:
    layers = [(256, 256, 4, 32), (256, 256, 8, 64), (128, 256, 16, 128), (64, 128, 32, 256)]
if len(sys.argv) > 3:   # explicit layers "ci,co,h,w;ci,co,h,w" (h, w = the gradient's size)
    layers = [tuple(int(v) for v in l.split(",")) for l in sys.argv[3].split(";")]
work = []
for ci, co, h, w in layers:
    x = torch.randn(n, ci, h * st, w * st, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    gy = torch.randn(n, co, h, w, device="cuda").to(torch.bfloat16).contiguous(memory_format=CL)
    gw = torch.zeros(3, 3, ci, co, device="cuda")
    gb = torch.zeros(co, device="cuda")
    work.append((x, gy, gw, gb))
flops = sum(2.0 * 9 * n * h * w * ci * co for ci, co, h, w in layers)
def run():
    K.defer_wgrad_reductions()
    for x, gy, gw, gb in work:
        K.conv2d_bwd_weight(x, gy, 3, st, 0.1, out=gw, bias_out=gb)
    K.flush_wgrad_reductions()
for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
best = 1e9
for _ in range(10):
    e0.record(); run(); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print("layers %s: " % (sys.argv[3] if len(sys.argv) > 3 else "default") + "stride %d, %d images: %.1f us per flush (group launch + fold), %.0f TFLOP/s" % (st, n, best * 1e3, flops / (best * 1e-3) / 1e12))
