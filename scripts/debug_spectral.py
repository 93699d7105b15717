import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import spectral_np as S
from gansynth_amd import spectral_ops as G
P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)
t = np.arange(64000) / 16000.0
tone = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 880.0 * t)
noise = np.clip(np.random.default_rng(4000).normal(0.0, 0.1, 64000), -1, 1)
w = np.stack([tone, noise]).astype(np.float32)
st = S.convert_to_spectrogram_stages(w, **P)
st64 = S.convert_to_spectrogram_stages(w, **P, dtype=np.float64)
lm, mi = G.convert_to_spectrogram(torch.from_numpy(w).cuda(), **P)
lm, mi = lm.cpu().numpy(), mi.cpu().numpy()
mm = np.exp(lm * 10.05 - 3.76) - 1e-6
for i, name in enumerate(["tone", "noise"]):
    d = np.abs(lm[i] - st["log_mel"][i]); d64 = np.abs(lm[i] - st64["log_mel"][i]); o = np.abs(st["log_mel"][i] - st64["log_mel"][i])
    print(name, "log_mel |gpu-o32| max", d.max(), "|gpu-o64|", d64.max(), "|o32-o64|", o.max())
    sc = st64["mel_magnitude"][i].max()
    print("   mel_mag lin rel err gpu-o64", np.abs(mm[i] - st64["mel_magnitude"][i]).max() / sc, "o32-o64", np.abs(st["mel_magnitude"][i] - st64["mel_magnitude"][i]).max() / sc)
    loud = st64["mel_magnitude"][i] > 1e-3 * sc
    print("   loud frac", loud.mean(), "log err on loud gpu-o64", d64[loud].max(), "o32-o64", o[loud].max())
    wr = lambda x: (x + 1) % 2 - 1
    e = np.abs(wr(mi[i] - st64["mel_if"][i])); eo = np.abs(wr(st["mel_if"][i] - st64["mel_if"][i]))
    print("   IF frac<1e-3 gpu-o64", np.mean(e < 1e-3), "o32-o64", np.mean(eo < 1e-3), " on loud:", np.mean(e[loud] < 1e-3), np.mean(eo[loud] < 1e-3))
