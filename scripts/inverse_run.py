"""(GPU) inverse spectral path, 256 examples: (log-mel, IF) images -> waveforms (spectral_ops.py:97-149). usage: inverse_run.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gansynth_amd import spectral_ops as G

P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
B = 256
rng = np.random.default_rng(4000)
x = torch.from_numpy(np.clip(rng.normal(0.0, 0.1, (B, 64000)), -1, 1).astype(np.float32)).cuda()
img = G.convert_to_images(x, **P)
for _ in range(100):   # (clock ramp: the first launches after idle run slow)
    wav = G.convert_images_to_waveform(img, **P)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(iters):
    wav = G.convert_images_to_waveform(img, **P)
e.record()
torch.cuda.synchronize()
us = s.elapsed_time(e) / iters * 1e3
print(f"inverse batch {B}: {us:.1f} us per batch; pinv contraction 2 x {2 * B * 128 * 1024 * 1024 / 1e9:.1f} GFLOP")
