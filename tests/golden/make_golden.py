"""Generates the golden fixtures under tests/golden/ from the CPU oracle (oracle/).

Run from the repo root:  python tests/golden/make_golden.py
The reference (TF 1.13) cannot be executed in this image, so these are
oracle-generated pins ("parity unpinned" w.r.t. real TensorFlow outputs), used to (a) freeze the
oracle against accidental edits and (b) check the HIP path on the GPU box.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import spectral_np as S  # noqa: E402
from oracle import torch_ref as R  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
SPECTRAL = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)


def tone_and_noise():
    t = np.arange(64000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 880.0 * t)
    noise = np.clip(np.random.default_rng(4000).normal(0.0, 0.1, 64000), -1, 1)
    return np.stack([tone, noise]).astype(np.float32)


def pggan_2x16():
    """BASELINE.json configs[0] with SURVEY.md D1/D2: lowest stage is 2x16, batch must be 4."""
    pg = R.PGGAN([2, 16], [128, 1024], 32, 256, 0.0)
    gp, dp = pg.init_params(seed=0, bias_std=0.1)
    lat, lab, real = R.synthetic_batch(4, rank=0)
    fake = pg.generator(gp, lat, lab)
    feats, logits = pg.discriminator(dp, real, lab)
    tr = R.Trainer(pg, gp, dp)
    d_loss, d_grads = tr.d_step(lat, lab, real)
    lat2, lab2, _ = R.synthetic_batch(4, rank=1)
    g_loss, g_grads = tr.g_step(lat2, lab2)
    out = dict(
        fake_2x16=fake[:, :, ::64, ::64].detach().numpy(),  # growing_depth 0: images are the 2x16 stage upscaled x64
        features=feats.detach().numpy(), logits=logits.detach().numpy(),
        d_loss=np.float32(d_loss), g_loss=np.float32(g_loss),
    )
    for k, g in list(d_grads.items()) + list(g_grads.items()):
        out["gradnorm/" + k] = np.float32(g.double().norm())
    for k, p in list(tr.d.items()) + list(tr.g.items()):
        out["paramsum/" + k] = np.float64(p.detach().double().sum())
    np.savez_compressed(os.path.join(HERE, "pggan_2x16_b4.npz"), **out)
    print("pggan_2x16_b4: d_loss", float(d_loss), "g_loss", float(g_loss))


def sample_indices(numel, key, n=64):
    """n fixed pseudo-random flat indices (logical NCHW order) into a tensor of `numel` elements."""
    return np.random.default_rng(abs(hash_name(key)) % (2 ** 32)).integers(0, numel, size=n)


def hash_name(name):
    h = 2166136261
    for ch in name.encode():
        h = ((h ^ ch) * 16777619) % (2 ** 32)
    return h


def summarize(out, key, t):
    """Per-tensor pin: float64 sum, sum of |.|, and 64 sampled elements (SURVEY.md 8c pin 2)."""
    a = t.detach().double().flatten().numpy()
    out["sum/" + key] = np.float64(a.sum())
    out["sumabs/" + key] = np.float64(np.abs(a).sum())
    out["samples/" + key] = a[sample_indices(a.size, key)].astype(np.float32)
    out["numel/" + key] = np.int64(a.size)


def full_forward_tensors(pg, gp, dp, lat, lab, real):
    """{key: tensor} of one fully grown forward of both networks: the outputs and every leaky_relu OUTPUT in call order
    (generator, then discriminator on the real images)."""
    with torch.no_grad(), R.lrelu_tape("record") as rec:
        fake = pg.generator(gp, lat, lab)
        feats, logits = pg.discriminator(dp, real, lab)
    tensors = {"generator/images": fake, "discriminator/features": feats, "discriminator/logits": logits}
    for name, xs in rec.calls:
        for i, x in enumerate(xs):
            tensors["%s/leaky_relu_%02d" % (name, i)] = torch.nn.functional.leaky_relu(x, 0.2)
    return tensors


def pggan_full():
    """SURVEY.md 8(c) pin 2: one fully grown 128x1024x2 forward of G and D at batch 4 (BASELINE.json configs[1] shapes), every
    activation tensor as checksums + 64 sampled elements."""
    pg = R.PGGAN([2, 16], [128, 1024], 32, 256, 1.0)
    gp, dp = pg.init_params(seed=0, bias_std=0.1)
    lat, lab, real = R.synthetic_batch(4, rank=0)
    out = {}
    for key, t in full_forward_tensors(pg, gp, dp, lat, lab, real).items():
        summarize(out, key, t)
    np.savez_compressed(os.path.join(HERE, "pggan_full_b4.npz"), **out)
    print("pggan_full_b4:", len([k for k in out if k.startswith("sum/")]), "tensors")


def spectral():
    wave = tone_and_noise()
    st = S.convert_to_spectrogram_stages(wave, **SPECTRAL)
    frames = np.array([0, 3, 7, 40, 127])
    out = dict(frames=frames)
    for k in ("magnitude", "phase", "mel_magnitude", "mel_phase", "log_mel", "mel_if"):
        out[k] = st[k][:, frames, :]
    out["mel_nnz"] = np.int64((st["mel"] != 0).sum())
    out["mel_colsum"] = st["mel"].sum(0)
    wav = S.convert_to_waveform(st["log_mel"], st["mel_if"], **SPECTRAL)
    out["waveform_head"] = wav[:, 20000:20512]
    np.savez_compressed(os.path.join(HERE, "spectral_tone_noise.npz"), **out)
    print("spectral: log_mel range", st["log_mel"].min(), st["log_mel"].max())


if __name__ == "__main__":
    torch.manual_seed(0)
    pggan_2x16()
    pggan_full()
    spectral()
