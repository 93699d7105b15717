"""(GPU) BASELINE.json configs[3] runner for profilers: waveform -> (log-mel, IF) images, batch 256 x 64000 samples.
usage: spectral_run.py [iterations] [dtype f32|bf16] [batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gansynth_amd import spectral_ops as G

P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dtype = torch.bfloat16 if len(sys.argv) > 2 and sys.argv[2] == "bf16" else torch.float32
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(4000)
x = torch.from_numpy(np.clip(rng.normal(0.0, 0.1, (B, 64000)), -1, 1).astype(np.float32)).cuda()
for _ in range(3):
    img = G.convert_to_images(x, **P, dtype=dtype)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    img = G.convert_to_images(x, **P, dtype=dtype)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / iters * 1e3
print(f"{dtype} batch {B}: {us:.1f} us per batch, {B / us * 1e6:.0f} examples/s, algorithmic {B * (64000 * 4 + 2 * 128 * 1024 * img.element_size()) / us / 1e3:.0f} GB/s")
