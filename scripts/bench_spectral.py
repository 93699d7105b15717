"""BASELINE.json configs[3]: waveform -> (log-mel, IF), 256 x 64000 samples: HIP kernels vs the numpy oracle on the host."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gansynth_amd import spectral_ops as G
from oracle import spectral_np as S

P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)
B = 256
rng = np.random.default_rng(4000)
w = np.clip(rng.normal(0.0, 0.1, (B, 64000)), -1, 1).astype(np.float32)
x = torch.from_numpy(w).cuda()
for dtype in (torch.float32, torch.bfloat16):
    for _ in range(3):
        img = G.convert_to_images(x, **P, dtype=dtype)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20):
        img = G.convert_to_images(x, **P, dtype=dtype)
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) / 20 * 1e3
    alg = B * (64000 * 4 + 2 * 128 * 1024 * img.element_size())
    print(f"forward  {dtype}: {us:8.1f} us per batch of {B}  = {B / us * 1e6:10.0f} examples/s, algorithmic {alg / us / 1e3:7.0f} GB/s")
for _ in range(2):
    wav = G.convert_images_to_waveform(img.float(), **P)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(5):
    wav = G.convert_images_to_waveform(img.float(), **P)
torch.cuda.synchronize()
print(f"inverse  f32: {(time.time() - t0) / 5 * 1e6:8.1f} us per batch of {B}")
t0 = time.time()
S.convert_to_spectrogram(w[:32], **P)
dt = time.time() - t0
print(f"numpy oracle (1 thread, 32 examples): {32 / dt:8.1f} examples/s")
