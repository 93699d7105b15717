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
    """An IF difference modulo 2 (a +-pi branch flip is a 2.0 jump) -- only applied to bins PROVEN ill-conditioned, see
    if_conditioning."""
    return (d + 1.0) % 2.0 - 1.0


def if_conditioning(st64, margin=1e-3, cut=1e-4):
    """Where the reference's IF (spectral_ops.py:21-44) is discontinuous in its input, from the float64 oracle:
      on_cut[b,t,m]  the wrapped phase difference sits within `margin` rad of +-pi: wrap() may take either branch (IF = +-1);
      branch[b,t,m]  a linear bin feeding mel column m has |arg X| within `cut` of pi at frame t or t-1 (with magnitude): atan2
                     may return +pi or -pi there, which moves the mel phase by 2 pi w -- NOT a multiple of 2 pi.
    Everything else is well conditioned and must agree plainly."""
    ph = st64["mel_phase"]
    d = np.diff(ph, axis=-2)
    md = np.mod(d + np.pi, 2 * np.pi) - np.pi
    on_cut = np.zeros(ph.shape, bool)
    on_cut[:, 1:] = np.pi - np.abs(md) < margin
    near = near_branch(st64, cut)
    hit = (near.astype(np.float64) @ (st64["mel"] != 0).astype(np.float64)) > 0          # [b, t, m]
    branch = hit.copy()
    branch[:, 1:] |= hit[:, :-1]
    return on_cut, branch


def near_branch(st64, cut=1e-4):
    """[b, t, k]: linear bin k of frame t (with magnitude) has |arg X| within `cut` of pi -- atan2 may land on either side."""
    lin, mag = st64["phase"], st64["magnitude"]
    return (np.pi - np.abs(lin) < cut) & (mag > 1e-6 * mag.max())


def check_branch_bins(got, ref, st64, b, branch, where=None, tol=2e-3):
    """The bins check_if leaves out are not unchecked: where a linear bin k sits on the atan2 branch cut at frame t or t - 1, the mel
    phase of column m moves by +-2 pi w[k, m] (w = the mel weight) and IF = wrap(p[t] - p[t-1]) / pi by +-2 w[k, m] modulo 2.  Every
    such bin must equal the oracle's value up to a signed sum of those quanta over the (few) hit bins of its column."""
    import itertools
    where = np.ones(ref.shape, bool) if where is None else where
    near, mel = near_branch(st64)[b], st64["mel"]
    ts, ms = np.nonzero(branch & where)
    worst = 0.0
    for t, m in zip(ts, ms):
        ks = [k for k in np.nonzero(mel[:, m])[0] if near[t, k] or (t > 0 and near[t - 1, k])]
        quanta = [2.0 * float(mel[k, m]) for k in ks]
        # a bin on the cut at t AND t - 1 may flip at either frame or both: coefficients -2 .. 2 per hit bin (columns have <= 6 non-zeros)
        best = min(abs(float(wrap2(np.float64(got[t, m] - ref[t, m] - sum(c * q for c, q in zip(cs, quanta))))))
                   for cs in itertools.product((-2, -1, 0, 1, 2), repeat=len(quanta)))
        worst = max(worst, best)
        assert best < tol, (b, t, m, float(got[t, m]), float(ref[t, m]), quanta)
    return len(ts), worst


def check_if(got, ref, on_cut, branch, where=None, tol=1e-3, max_branch=2e-3):
    """IF parity: plain on the well-conditioned bins, modulo 2 on the branch cut of wrap(), nothing on atan2 branch hits; the
    ill-conditioned sets must stay the small sets they are."""
    where = np.ones(ref.shape, bool) if where is None else where
    plain = where & ~on_cut & ~branch
    assert np.abs(got - ref)[plain].max() < tol, np.abs(got - ref)[plain].max()
    cut = where & on_cut & ~branch
    if cut.any():
        assert np.abs(wrap2(got - ref))[cut].max() < tol
    assert on_cut[where].mean() < 2e-3 and branch[where].mean() < max_branch, (on_cut[where].mean(), branch[where].mean())


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
    assert np.abs(mi - st["mel_if"]).max() < 1e-5, np.abs(mi - st["mel_if"]).max()   # same inputs, same fp32 recurrence: no modulo


def test_fused_vs_oracle_and_golden():
    """Conditioning: log(mel + 1e-6) amplifies fp32 FFT round-off without bound where a bin is (numerically)
    silent -- the oracle's own float32 and float64 evaluations of the pure tone differ by 0.09 there -- and the
    phase of a silent bin is noise.  So: the noise example (dense spectrum) is compared everywhere at 1e-3;
    the tone example is compared in the linear mel domain everywhere (relative to the frame maximum) and in
    the log / IF domain on the bins that carry signal (> 1e-3 of the maximum)."""
    from gansynth_amd import spectral_ops as G
    w = waves()
    st = S.convert_to_spectrogram_stages(w, **P)
    st64 = S.convert_to_spectrogram_stages(w, **P, dtype=np.float64)
    lm, mi = G.convert_to_spectrogram(torch.from_numpy(w).cuda(), **P)
    lm, mi = lm.cpu().numpy(), mi.cpu().numpy()
    assert lm.shape == mi.shape == (2, 128, 1024)
    on_cut, branch = if_conditioning(st64)
    # noise: everywhere
    assert np.abs(lm[1] - st["log_mel"][1]).max() < 1e-3
    assert np.abs(lm[1] - st64["log_mel"][1]).max() < 1e-3
    check_if(mi[1], st64["mel_if"][1], on_cut[1], branch[1])
    check_if(mi[1], st["mel_if"][1], on_cut[1], branch[1])
    # tone: linear domain everywhere, log/IF where there is signal
    mel_lin = np.exp(lm * 10.05 - 3.76) - 1e-6
    for i in range(2):
        ref = st64["mel_magnitude"][i]
        assert np.abs(mel_lin[i] - ref).max() <= 3e-4 * ref.max()
        loud = ref > 1e-3 * ref.max()
        assert np.abs(lm[i] - st64["log_mel"][i])[loud].max() < 1e-3
        prev_loud = loud.copy()
        prev_loud[1:] &= loud[:-1]                     # IF at t is a difference of the phases at t and t - 1
        # (a stationary tone keeps re-visiting the same phases: ~1 % of its audible bins sit on the atan2 branch, 3e-5 of the noise's)
        check_if(mi[i], st64["mel_if"][i], on_cut[i], branch[i], where=prev_loud, max_branch=3e-2 if i == 0 else 2e-3)
        n_branch, _ = check_branch_bins(mi[i], st64["mel_if"][i], st64, i, branch[i] & ~on_cut[i], where=prev_loud)
        assert i == 1 or n_branch > 0   # (the stationary tone does visit the cut: the check above ran)
    assert np.allclose(lm[:, :3], (np.log(1e-6) + 3.76) / 10.05, atol=1e-6) and np.all(mi[:, :3] == 0)
    gold = np.load(os.path.join(GOLD, "spectral_tone_noise.npz"))
    fr = gold["frames"]
    assert np.abs(lm[1][fr] - gold["log_mel"][1]).max() < 1e-3
    check_if(mi[1][fr], gold["mel_if"][1], on_cut[1][fr], branch[1][fr])
    loud = gold["mel_magnitude"][0] > 1e-3 * gold["mel_magnitude"][0].max()
    assert np.abs(lm[0][fr] - gold["log_mel"][0])[loud].max() < 1e-3


def test_inverse_vs_oracle():
    from gansynth_amd import spectral_ops as G
    w = waves()
    lm, mi = S.convert_to_spectrogram(w, **P)
    ref32 = S.convert_to_waveform(lm, mi, **P)
    mel32 = S.linear_to_mel_weight_matrix(1024, 1024, 16000, 0.0, 8000.0, np.float32)
    ref64 = S.convert_to_waveform(lm.astype(np.float64), mi.astype(np.float64), **P, dtype=np.float64, mel_inverse=S.pinv(mel32))
    got = G.convert_to_waveform(torch.from_numpy(lm).cuda(), torch.from_numpy(mi).cuda(), **P).cpu().numpy()
    assert got.shape == ref32.shape == (2, 64000)
    # The contract is 1e-3 relative; measured (MI355X, round 2): 2.7e-6 of the peak against the exact (float64) evaluation with the
    # same float32-built pinv(mel), the float32 oracle itself sitting 2.7e-6 from it.  (Phases reach ~1e3 rad before cos / sin, so
    # the comparison must share the float32-built matrix: a float64-built pinv differs by 1.6e-4 relative = 0.16 rad.)
    for a, b32, b64 in zip(got, ref32, ref64):
        scale = np.abs(b64).max()
        own = np.abs(b32 - b64).max() / scale            # what float32 arithmetic costs the reference itself
        err = np.abs(a - b64).max() / scale
        print(f"inverse: HIP vs exact {err:.2e}, fp32 oracle vs exact {own:.2e}")
        assert err < 1e-4 and np.abs(a - b32).max() / scale < 1e-4, (err, own)
        assert S.cross_correlation(a, b64) > 0.999999


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
    # the IF of two rows against the oracle as in test_fused_vs_oracle_and_golden (at this batch a block is one example: twelve runs
    # exchanging their edge phases inside the block), and the last rows of the batch against the same rows alone
    st64 = S.convert_to_spectrogram_stages(w[:2], **P, dtype=np.float64)
    on_cut, branch = if_conditioning(st64)
    for i in range(2):
        check_if(img[i, 1].cpu().numpy(), st64["mel_if"][i], on_cut[i], branch[i])
    tail = G.convert_to_images(x[254:256], **P)
    assert torch.equal(tail, img[254:256])


def test_inverse_batch256_properties():
    """The inverse path at configs[3]'s batch: 256 image pairs -> waveforms through the split-bf16 pinv contraction and the
    wave-per-frame inverse STFT with the overlap-add inside (block = example, neighbouring runs exchange their window tails).
    Examples are independent (rows of the batch == the same rows alone, to the bit), the round trip waveform -> images -> waveform
    returns the input on the un-padded interior, and one row matches the numpy oracle."""
    from gansynth_amd import spectral_ops as G
    rng = np.random.default_rng(4001)
    w = np.clip(rng.normal(0.0, 0.1, (256, 64000)), -1, 1).astype(np.float32)
    x = torch.from_numpy(w).cuda()
    img = G.convert_to_images(x, **P)
    wav = G.convert_images_to_waveform(img, **P)
    assert wav.shape == (256, 64000) and torch.isfinite(wav).all()
    sub = G.convert_images_to_waveform(img[100:102].contiguous(), **P)
    assert torch.equal(sub, wav[100:102])
    lm, mi = img[3:4, 0].cpu().numpy(), img[3:4, 1].cpu().numpy()
    ref = S.convert_to_waveform(lm, mi, **P)[0]
    got = wav[3].cpu().numpy()
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-3
    assert S.cross_correlation(got, ref) > 0.99999
