"""GPU: waveform <-> (log-mel, IF) HIP kernels vs the numpy oracle (stage-wise and fused), the
committed golden rows, and batch-256 size-independent properties (BASELINE.json configs[3])."""
import os

import numpy as np
import pytest
import torch

from oracle import spectral_np as S

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)


def waves():
    t = np.arange(64000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 880.0 * t)
    noise = np.clip(np.random.default_rng(4000).normal(0.0, 0.1, 64000), -1, 1)
    return np.stack([tone, noise]).astype(np.float32)


def wrap2(d):
    """IF differences are compared modulo 2 (a +-pi branch flip of an ill-conditioned phase is a 2.0 jump)."""
    return (d + 1.0) % 2.0 - 1.0


def test_stagewise_vs_oracle():
    from gansynth_amd import spectral_ops as G
    w = waves()
    st = S.convert_to_spectrogram_stages(w, **P)
    mag, ph = G.stft_magnitude_phase(torch.from_numpy(w).cuda(), **P)
    mag, ph = mag.cpu().numpy(), ph.cpu().numpy()
    scale = st["magnitude"].max(axis=(1, 2), keepdims=True)
    assert np.abs(mag - st["magnitude"]).max() / scale.max() < 1e-5
    strong = st["magnitude"] > 1e-3 * scale  # phase is only defined where there is signal
    dphi = np.angle(np.exp(1j * (ph - st["phase"])))
    assert np.abs(dphi[strong]).max() < 1e-3
    assert np.all(ph[:, :3] == 0) and np.all(mag[:, :3] == 0)  # frames 0-2 are pure front padding
    # mel projection and IF on the oracle's own intermediates (isolates each kernel)
    mm = G.mel_project(torch.from_numpy(st["magnitude"]).cuda(), **P).cpu().numpy()
    assert np.abs(mm - st["mel_magnitude"]).max() <= 1e-5 * st["mel_magnitude"].max()
    mi = G.instantaneous_frequency(torch.from_numpy(st["mel_phase"]).cuda(), **P).cpu().numpy()
    d = wrap2(mi - st["mel_if"])
    assert np.abs(d).max() < 1e-3, np.abs(d).max()


def test_fused_vs_oracle_and_golden():
    from gansynth_amd import spectral_ops as G
    w = waves()
    st = S.convert_to_spectrogram_stages(w, **P)
    lm, mi = G.convert_to_spectrogram(torch.from_numpy(w).cuda(), **P)
    lm, mi = lm.cpu().numpy(), mi.cpu().numpy()
    assert lm.shape == mi.shape == (2, 128, 1024)
    assert np.abs(lm - st["log_mel"]).max() < 1e-3
    d = np.abs(wrap2(mi - st["mel_if"]))
    # the mel-projected phase mixes up to 6 raw phases; elements fed by near-silent bins are ill-conditioned
    assert np.mean(d < 1e-3) > 0.995, np.mean(d < 1e-3)
    assert np.allclose(lm[:, :3], (np.log(1e-6) + 3.76) / 10.05, atol=1e-6) and np.all(mi[:, :3] == 0)
    gold = np.load(os.path.join(GOLD, "spectral_tone_noise.npz"))
    fr = gold["frames"]
    assert np.abs(lm[:, fr] - gold["log_mel"]).max() < 1e-3
    assert np.mean(np.abs(wrap2(mi[:, fr] - gold["mel_if"])) < 1e-3) > 0.995


def test_inverse_vs_oracle():
    from gansynth_amd import spectral_ops as G
    w = waves()
    lm, mi = S.convert_to_spectrogram(w, **P)
    ref = S.convert_to_waveform(lm, mi, **P)
    got = G.convert_to_waveform(torch.from_numpy(lm).cuda(), torch.from_numpy(mi).cuda(), **P).cpu().numpy()
    assert got.shape == ref.shape == (2, 64000)
    # cos/sin of phases up to ~1e3 rad amplify fp32 rounding of the pinv contraction: compare by correlation and rms
    for a, b in zip(got, ref):
        assert S.cross_correlation(a, b) > 0.999
        assert np.sqrt(np.mean((a - b) ** 2)) < 2e-2 * np.sqrt(np.mean(b ** 2))


def test_batch256_properties():
    """configs[3]: 256 x 64000 samples.  Examples are independent (row i of a batch == the same row alone),
    padding frames are exact, outputs finite, IF within [-1, 1]."""
    from gansynth_amd import spectral_ops as G
    rng = np.random.default_rng(4000)
    w = np.clip(rng.normal(0.0, 0.1, (256, 64000)), -1, 1).astype(np.float32)
    x = torch.from_numpy(w).cuda()
    img = G.convert_to_images(x, **P)
    assert img.shape == (256, 2, 128, 1024)
    assert torch.isfinite(img).all()
    assert float(img[:, 1].abs().max()) <= 1.0 + 1e-4
    sub = G.convert_to_images(x[17:19], **P)
    assert torch.equal(sub, img[17:19])
    assert torch.all(img[:, 1, :3] == 0)
    ref_lm, _ = S.convert_to_spectrogram(w[:2], **P)
    assert np.abs(img[:2, 0].cpu().numpy() - ref_lm).max() < 1e-3
