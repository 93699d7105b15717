"""What a kernel NODE costs in a replayed hipGraph on this stack, by what the node is: the SAME tiny kernel repeated, a rotation of
DISTINCT tiny torch kernels, a rotation of the library's own tiny kernels (pixel norm on 16 pixels, act_bwd, axpby, bias_act ...), and
the same with 20-us kernels in between.  (rocprofv3 shows ~4.7 us for every tiny kernel of the captured training step where a chain of
identical trivial kernels costs ~1.5 us per node: which property of the step's kernels is it?)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gansynth_amd import kernels

K = kernels.get()
dev = torch.device("cuda:0")
CL = torch.channels_last
x = torch.zeros(64, device=dev)
y = torch.zeros(64, device=dev)
z = torch.randn(8, 256, 2, 16, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
g = torch.randn(8, 256, 2, 16, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
bias = torch.zeros(256, device=dev)
big = torch.randn(8, 256, 16, 128, device=dev).to(torch.bfloat16).contiguous(memory_format=CL)
wt = torch.randn(3, 3, 256, 256, device=dev)


def bench(fns, n, label):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for f in fns:
            f()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for i in range(n):
            fns[i % len(fns)]()
    for _ in range(3):
        gr.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10):
            gr.replay()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    print(f"{label}: {best * 1e6 / n:.2f} us per node ({n} nodes, {len(fns)} distinct)")
    return best / n


torch_ops = [lambda: x.add_(1.0), lambda: x.mul_(0.5), lambda: x.neg_(), lambda: x.abs_(), lambda: x.tanh_(), lambda: x.sigmoid_(),
             lambda: x.sub_(y), lambda: x.exp_(), lambda: x.clamp_(-1, 1), lambda: x.sqrt_(), lambda: x.copy_(y), lambda: x.fill_(0.5),
             lambda: x.floor_(), lambda: x.sin_(), lambda: x.cos_(), lambda: x.reciprocal_()]
bench(torch_ops[:1], 512, "torch, one tiny kernel")
bench(torch_ops[:2], 512, "torch, 2 tiny kernels alternating")
bench(torch_ops[:4], 512, "torch, 4 distinct tiny kernels")
bench(torch_ops, 512, "torch, 16 distinct tiny kernels")
lib_ops = [lambda: K.pixel_norm_fwd(z, 1e-8), lambda: K.act_bwd(g, z, 1), lambda: K.axpby(z, g, 1.0, 1.0), lambda: K.bias_act_fwd(z, bias, 1),
           lambda: K.pixel_norm_bwd(g, z, 1e-8), lambda: K.channel_sum(g)]
bench(lib_ops[:1], 240, "library, pixel_norm_fwd on [8,256,2,16] repeated")
bench(lib_ops, 240, "library, 6 distinct tiny ops (7-8 kernels) rotating")
conv = lambda: K.conv2d_fwd_bias_act(big, wt, bias, 3, 1, 0.05, 1)
conv()
t_conv = bench([conv], 100, "library, 256->256 conv @16x128 x8 repeated")
t_mix = bench([conv, lambda: K.pixel_norm_fwd(z, 1e-8)], 200, "library, that conv alternating with the tiny pixel_norm")
print(f"  => the tiny kernel behind a 20-us conv costs {(2 * t_mix - t_conv) * 1e6:.2f} us")
t_mix = bench([conv, lambda: x.add_(1.0)], 200, "that conv alternating with torch's tiny add_")
print(f"  => torch's tiny add_ behind a 20-us conv costs {(2 * t_mix - t_conv) * 1e6:.2f} us")
