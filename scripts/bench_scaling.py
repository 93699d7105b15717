import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gansynth_amd import kernels
K = kernels.get()
dt = torch.bfloat16
def run(n, ci, co, h, w):
    x = torch.randn(n, ci, h, w, device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(3, 3, ci, co, device="cuda")
    for _ in range(6):
        K.conv2d_fwd(x, wt, 3, 1, 0.1)
    torch.cuda.synchronize()
for ci in (32, 64, 128, 256, 512):
    run(8, ci, 64, 16, 128)      # 64 tiles of 256 px -> 64 blocks, stages = ci/32
for h in (16, 32, 64, 128):
    run(8, 64, 64, h, 128)       # more tiles at fixed K
