"""CPU: pins for the oracle itself.  The reference ships no tests or golden vectors, so the
oracle is pinned by hand-derivable known answers, by two independent restatements agreeing (numpy direct
definition vs torch library ops) and by the committed fixtures it generated (frozen against edits)."""
import os

import numpy as np
import torch

from oracle import ops_np as N
from oracle import spectral_np as S
from oracle import torch_ref as R

GOLD = os.path.join(os.path.dirname(__file__), "golden")
P = dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75)


def test_exact_index_ops():
    x = np.arange(2 * 3 * 2 * 4, dtype=np.float64).reshape(2, 3, 2, 4)
    up = N.upscale2d(x, (2, 3))
    assert up.shape == (2, 3, 4, 12) and up[1, 2, 3, 11] == x[1, 2, 1, 3] and up[0, 0, 1, 2] == x[0, 0, 0, 0]
    assert np.array_equal(N.downscale2d(up, (2, 3)), x)  # mean of a constant block
    lab = np.eye(5)[[3, 0, 4]]
    w = np.arange(20.0).reshape(5, 4)
    assert np.array_equal(N.embedding(lab, w, 1.0), w[[3, 0, 4]] * np.sqrt(1.0 / 5))


def test_same_padding_asymmetry_and_transpose_crop():
    """stride-2 SAME on an even input pads 0 before / 1 after; conv2d_transpose is its input-gradient."""
    x = np.zeros((1, 1, 4, 4)); x[0, 0, 0, 0] = 1.0
    w = np.arange(1.0, 10.0).reshape(3, 3, 1, 1) / N.weight_scale((3, 3, 1, 1), 2.0)
    y = N.conv2d(x, w, np.zeros(1), (2, 2))
    assert y.shape == (1, 1, 2, 2) and abs(y[0, 0, 0, 0] - 1.0) < 1e-12 and np.count_nonzero(y) == 1  # only tap (0,0) sees pixel (0,0)
    z = np.zeros((1, 1, 2, 2)); z[0, 0, 1, 1] = 1.0
    t = N.conv2d_transpose(z, w, np.zeros(1))
    assert t.shape == (1, 1, 4, 4)
    assert np.allclose(t[0, 0, 2:, 2:], [[1, 2], [4, 5]])  # taps (2,*) and (*,2) fall off the cropped end
    assert np.count_nonzero(t[0, 0, :2]) == 0


def test_numpy_and_torch_restatements_agree():
    rng = np.random.default_rng(0)
    x = rng.standard_normal((4, 5, 6, 8)); w = rng.standard_normal((3, 3, 5, 7)); b = rng.standard_normal(7)
    tx, tw, tb = map(torch.tensor, (x, w, b))
    for s in [(1, 1), (2, 2)]:
        assert np.abs(N.conv2d(x, w, b, s) - R.conv2d(tx, tw, tb, s).numpy()).max() < 1e-12
    assert np.abs(N.conv2d_transpose(x, w, b) - R.conv2d_transpose(tx, tw, tb).numpy()).max() < 1e-12
    assert np.abs(N.batch_stddev(x) - R.batch_stddev(tx).numpy()).max() < 1e-12
    assert np.abs(N.pixel_normalization(x) - R.pixel_normalization(tx).numpy()).max() < 1e-12
    w1 = rng.standard_normal((1, 1, 5, 2))
    assert np.abs(N.conv2d(x, w1, b[:2]) - R.conv2d(tx, torch.tensor(w1), tb[:2]).numpy()).max() < 1e-12


def test_batch_stddev_group_layout():
    x = np.zeros((8, 1, 1, 1)); x[[0, 2, 4, 6], 0, 0, 0] = [1.0, 3.0, 5.0, 7.0]  # column 0 = samples 0,2,4,6
    y = N.batch_stddev(x, epsilon=0.0)
    assert np.allclose(y[[0, 2, 4, 6]], np.std([1.0, 3.0, 5.0, 7.0])) and np.allclose(y[[1, 3, 5, 7]], 0.0)


def test_tf_adam_form():
    p = {"w": torch.tensor([1.0])}; g = {"w": torch.tensor([0.5])}
    m = {"w": torch.zeros(1)}; v = {"w": torch.zeros(1)}
    R.adam_tf_step(p, g, m, v, 1, 0.1, 0.0, 0.99, 1e-8)
    lr_t = 0.1 * np.sqrt(1 - 0.99)
    assert np.isclose(float(p["w"]), 1.0 - lr_t * 0.5 / (np.sqrt(0.01 * 0.25) + 1e-8), rtol=1e-6)


def test_growth_schedule():
    pg = R.PGGAN([2, 16], [128, 1024], 32, 256, 0.0)
    assert pg.max_depth == 6 and pg.growing_depth == 0.0
    assert [pg.channels(d) for d in range(-1, 7)] == [256, 256, 256, 256, 256, 128, 64, 32]
    pg.growing_level = 1.0
    assert abs(pg.growing_depth - 7.0) < 1e-6
    g, d = pg.variable_shapes()
    assert sum(int(np.prod(s)) for s in g.values()) == 8932238 and sum(int(np.prod(s)) for s in d.values()) == 6830973


def test_hann_and_mel_structure():
    w = S.hann_window(2048, np.float64)
    assert w[0] == 0.0 and abs(w[1024] - 1.0) < 1e-15
    assert np.allclose((w ** 2).reshape(4, 512).sum(0), 1.5)
    mel = S.linear_to_mel_weight_matrix(1024, 1024, 16000, 0.0, 8000.0, np.float32)
    assert mel.shape == (1024, 1024) and (mel != 0).sum() == 2042
    assert (mel != 0).sum(0).max() == 6 and (mel != 0).sum(1).max() == 2 and ((mel != 0).sum(0) == 0).sum() == 107
    assert np.all(mel[0] == 0)


def test_stft_sinusoid_peak_and_zero_frames():
    t = np.arange(64000)
    k = 64  # bin-centred: f = k * 16000 / 2048
    x = np.cos(2 * np.pi * k * t / 2048.0)[None].astype(np.float32)
    st = S.convert_to_spectrogram_stages(x, **P)
    assert np.all(st["magnitude"][0, :3] == 0) and np.all(st["phase"][0, :3] == 0)
    assert np.argmax(st["magnitude"][0, 64]) == k - 1  # DC dropped
    assert abs(st["magnitude"][0, 64, k - 1] - 512.0) < 1e-2  # sum(hann)/2
    assert np.allclose(st["log_mel"][0, :3], (np.log(1e-6) + 3.76) / 10.05, atol=1e-6)
    assert np.all(st["mel_if"][0, :3] == 0)


def test_instantaneous_frequency_of_linear_ramp():
    om = 2.5
    t = np.arange(16, dtype=np.float64)
    ph = np.angle(np.exp(1j * (0.3 + om * t)))[None, :, None]
    f = S.instantaneous_frequency(ph, axis=-2)
    assert abs(f[0, 0, 0] - 0.3 / np.pi) < 1e-12
    assert np.allclose(f[0, 1:, 0], om / np.pi, atol=1e-9)


def test_istft_inverts_stft():
    x = np.random.default_rng(1).standard_normal((1, 67072)).astype(np.float32)
    y = S.inverse_stft(S.stft(x, 2048, 512), 2048, 512)
    assert np.abs(y[:, 2048:-2048] - x[:, 2048:-2048]).max() < 1e-5


def test_golden_fixtures_are_current():
    gold = np.load(os.path.join(GOLD, "spectral_tone_noise.npz"))
    t = np.arange(64000) / 16000.0
    tone = 0.5 * np.sin(2 * np.pi * 440.0 * t) + 0.25 * np.sin(2 * np.pi * 880.0 * t)
    noise = np.clip(np.random.default_rng(4000).normal(0.0, 0.1, 64000), -1, 1)
    st = S.convert_to_spectrogram_stages(np.stack([tone, noise]).astype(np.float32), **P)
    fr = gold["frames"]
    assert np.abs(st["log_mel"][:, fr] - gold["log_mel"]).max() < 1e-5
    assert np.mean(np.abs(st["mel_if"][:, fr] - gold["mel_if"]) < 1e-4) > 0.999
    g2 = np.load(os.path.join(GOLD, "pggan_2x16_b4.npz"))
    pg = R.PGGAN([2, 16], [128, 1024], 32, 256, 0.0)
    gp, dp = pg.init_params(seed=0, bias_std=0.1)
    lat, lab, real = R.synthetic_batch(4, rank=0)
    fake = pg.generator(gp, lat, lab)
    assert np.abs(fake[:, :, ::64, ::64].detach().numpy() - g2["fake_2x16"]).max() < 1e-5


def test_full_size_forward_fixture_is_current():
    """SURVEY.md 8(c) pin 2: the fully grown 128x1024x2 forward of both networks at batch 4 -- the outputs and every leaky_relu
    output, as float64 checksums + 64 sampled elements per tensor (tests/golden/pggan_full_b4.npz) -- freezes the full-size oracle
    against edits.  (The GPU suite checks the HIP path against the same file.)"""
    from tests.golden import make_golden as MG
    gold = np.load(os.path.join(GOLD, "pggan_full_b4.npz"))
    pg = R.PGGAN([2, 16], [128, 1024], 32, 256, 1.0)
    gp, dp = pg.init_params(seed=0, bias_std=0.1)
    lat, lab, real = R.synthetic_batch(4, rank=0)
    tensors = MG.full_forward_tensors(pg, gp, dp, lat, lab, real)
    keys = sorted(k[len("sum/"):] for k in gold.files if k.startswith("sum/"))
    assert keys == sorted(tensors) and len(keys) == 32
    assert tuple(tensors["generator/images"].shape) == (4, 2, 128, 1024)
    for key in keys:
        a = tensors[key].detach().double().flatten().numpy()
        assert a.size == int(gold["numel/" + key])
        assert abs(a.sum() - float(gold["sum/" + key])) <= 1e-6 * float(gold["sumabs/" + key]), key
        assert abs(np.abs(a).sum() - float(gold["sumabs/" + key])) <= 1e-6 * float(gold["sumabs/" + key]), key
        got = a[MG.sample_indices(a.size, key)]
        assert np.abs(got - gold["samples/" + key]).max() <= 1e-5 * max(1.0, np.abs(gold["samples/" + key]).max()), key

