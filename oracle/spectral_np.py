"""numpy restatement of reference spectral_ops.py:8-149 (waveform <-> log-mel + IF).

Test infrastructure (see oracle/__init__.py).  `dtype` selects the working
precision: float32 follows TF's arithmetic type (the mel matrix in particular is
built in float32 by tf.signal.linear_to_mel_weight_matrix in TF 1.13), float64 is
the high-precision twin used to bound rounding error in tests.
"""
import numpy as np

_MEL_BREAK_FREQUENCY_HERTZ = 700.0
_MEL_HIGH_FREQUENCY_Q = 1127.0


def _params(spectrogram_shape, overlap):
    """spectral_ops.py:50-53."""
    time_steps, num_freq_bins = spectrogram_shape
    frame_length = num_freq_bins * 2
    frame_step = int((1.0 - overlap) * frame_length)
    num_samples = frame_step * (time_steps - 1) + frame_length
    return time_steps, num_freq_bins, frame_length, frame_step, num_samples


def hann_window(n, dtype=np.float32):
    """tf.signal.hann_window(periodic=True): 0.5 - 0.5 cos(2 pi k / n)."""
    k = np.arange(n, dtype=np.float64)
    return (0.5 - 0.5 * np.cos(2.0 * np.pi * k / n)).astype(dtype)


def hertz_to_mel(f, dtype):
    f = np.asarray(f, dtype=dtype)
    return (dtype(_MEL_HIGH_FREQUENCY_Q) * np.log(dtype(1.0) + f / dtype(_MEL_BREAK_FREQUENCY_HERTZ))).astype(dtype)


def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate,
                                lower_edge_hertz, upper_edge_hertz, dtype=np.float32):
    """tf.signal.linear_to_mel_weight_matrix (TF 1.13 mel_ops.py), called at
    spectral_ops.py:76-82: HTK mel, DC bin excluded then re-added as a zero row."""
    dtype = np.dtype(dtype).type
    bands_to_zero = 1
    nyquist = dtype(sample_rate / 2.0)
    lin = np.linspace(dtype(0.0), nyquist, num_spectrogram_bins, dtype=dtype)[bands_to_zero:]
    spec_mel = hertz_to_mel(lin, dtype)[:, None]
    edges = np.linspace(hertz_to_mel(lower_edge_hertz, dtype), hertz_to_mel(upper_edge_hertz, dtype),
                        num_mel_bins + 2, dtype=dtype)
    lower, center, upper = edges[:-2][None, :], edges[1:-1][None, :], edges[2:][None, :]
    lower_slopes = (spec_mel - lower) / (center - lower)
    upper_slopes = (upper - spec_mel) / (upper - center)
    w = np.maximum(dtype(0.0), np.minimum(lower_slopes, upper_slopes))
    return np.pad(w, [[bands_to_zero, 0], [0, 0]]).astype(dtype)


def diff(x, axis):
    """spectral_ops.py:8-18."""
    a = np.swapaxes(x, axis, -1)
    return np.swapaxes(a[..., 1:] - a[..., :-1], axis, -1)


def unwrap(phases, axis=-1):
    """spectral_ops.py:21-33 (tf.mod is floor-mod; np.pi enters as a dtype constant)."""
    dt = phases.dtype.type
    pi = dt(np.float32(np.pi)) if phases.dtype == np.float32 else dt(np.pi)
    diffs = diff(phases, axis)
    mods = np.mod(diffs + pi, pi * dt(2.0)) - pi
    idx = np.logical_and(mods == -pi, diffs > 0)
    mods = np.where(idx, pi, mods)
    corrects = mods - diffs
    cums = np.cumsum(corrects, axis=axis, dtype=phases.dtype)
    shape = list(phases.shape)
    shape[axis] = 1
    cums = np.concatenate([np.zeros(shape, dtype=phases.dtype), cums], axis=axis)
    return phases + cums


def instantaneous_frequency(phases, axis=-2):
    """spectral_ops.py:36-44."""
    dt = phases.dtype.type
    pi = dt(np.float32(np.pi)) if phases.dtype == np.float32 else dt(np.pi)
    un = unwrap(phases, axis=axis)
    d = diff(un, axis)
    init = np.take(un, [0], axis=axis)
    return np.concatenate([init, d], axis=axis) / pi


def stft(waveforms, frame_length, frame_step, dtype=np.float32):
    """tf.signal.stft(pad_end=False, fft_length=frame_length (power of 2), periodic Hann):
    frames x[step*i : step*i+L] * w -> rfft."""
    x = np.asarray(waveforms, dtype=dtype)
    n = (x.shape[-1] - frame_length) // frame_step + 1
    idx = np.arange(frame_length)[None, :] + frame_step * np.arange(n)[:, None]
    frames = x[..., idx] * hann_window(frame_length, dtype)
    out = np.fft.rfft(frames, n=frame_length, axis=-1)
    return out.astype(np.complex64 if dtype == np.float32 else np.complex128)


def convert_to_spectrogram_stages(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap,
                                  dtype=np.float32):
    """spectral_ops.py:45-94, returning every intermediate for stage-wise parity."""
    dt = np.dtype(dtype).type
    time_steps, nbins, frame_length, frame_step, num_samples = _params(spectrogram_shape, overlap)
    x = np.asarray(waveforms, dtype=dtype)
    x = np.pad(x, [[0, 0], [num_samples - waveform_length, 0]])          # :57 all padding in FRONT
    s = stft(x, frame_length, frame_step, dtype)[..., 1:]                 # :59-69 discard DC
    mag = np.abs(s).astype(dtype)                                         # :71
    phase = np.angle(s).astype(dtype)                                     # :72
    mel = linear_to_mel_weight_matrix(nbins, nbins, sample_rate, 0.0, sample_rate / 2.0, dtype)
    mel_mag = (mag @ mel).astype(dtype)                                   # :83
    mel_phase = (phase @ mel).astype(dtype)                               # :85
    log_mel = np.log(mel_mag + dt(1.0e-6))                                # :88
    mel_if = instantaneous_frequency(mel_phase, axis=-2)                  # :89
    log_mel = (log_mel - dt(-3.76)) / dt(10.05)                           # :91
    mel_if = (mel_if - dt(0.0)) / dt(1.0)                                 # :92
    return dict(stft=s, magnitude=mag, phase=phase, mel=mel, mel_magnitude=mel_mag, mel_phase=mel_phase,
                log_mel=log_mel.astype(dtype), mel_if=mel_if.astype(dtype))


def convert_to_spectrogram(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap, dtype=np.float32):
    st = convert_to_spectrogram_stages(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap, dtype)
    return st["log_mel"], st["mel_if"]


def pinv(a, rcond=None):
    """tfp.math.pinv (spectral_ops.py:122): SVD, singular values <= rcond*max dropped,
    default rcond = 10 * max(rows, cols) * eps(dtype)."""
    a = np.asarray(a)
    if rcond is None:
        rcond = 10.0 * max(a.shape[-2:]) * np.finfo(a.dtype).eps
    u, s, vt = np.linalg.svd(a.astype(np.float64), full_matrices=False)
    cutoff = rcond * s.max()
    sinv = np.where(s > cutoff, 1.0 / np.where(s > cutoff, s, 1.0), 0.0)
    return ((vt.T * sinv) @ u.T).astype(a.dtype)


def inverse_stft_window(frame_length, frame_step, dtype=np.float32):
    """tf.signal.inverse_stft_window_fn(frame_step, hann periodic): w / sum_overlaps w^2."""
    w = hann_window(frame_length, np.float64)
    overlaps = -(-frame_length // frame_step)
    denom = np.pad(w * w, (0, overlaps * frame_step - frame_length)).reshape(overlaps, frame_step).sum(0, keepdims=True)
    denom = np.tile(denom, (overlaps, 1)).reshape(-1)[:frame_length]
    return (w / denom).astype(dtype)


def inverse_stft(stfts, frame_length, frame_step, dtype=np.float32):
    """tf.signal.inverse_stft: irfft -> [:frame_length] * window -> overlap_and_add."""
    frames = np.fft.irfft(stfts, n=frame_length, axis=-1)[..., :frame_length].astype(dtype)
    frames = frames * inverse_stft_window(frame_length, frame_step, dtype)
    n = frames.shape[-2]
    out = np.zeros(frames.shape[:-2] + (frame_step * (n - 1) + frame_length,), dtype=dtype)
    for i in range(n):
        out[..., i * frame_step:i * frame_step + frame_length] += frames[..., i, :]
    return out


def convert_to_waveform(log_mel, mel_if, waveform_length, sample_rate, spectrogram_shape, overlap, dtype=np.float32,
                        mel_inverse=None):
    """spectral_ops.py:97-149.  `mel_inverse`: use this pinv(mel) (cast to `dtype`) instead of building one in `dtype` -- the
    float64 twin of a float32 run must contract with the SAME float32-built matrix to isolate arithmetic error."""
    dt = np.dtype(dtype).type
    pi = dt(np.float32(np.pi)) if dtype == np.float32 else dt(np.pi)
    time_steps, nbins, frame_length, frame_step, num_samples = _params(spectrogram_shape, overlap)
    log_mel = np.asarray(log_mel, dtype=dtype) * dt(10.05) + dt(-3.76)    # :107
    mel_if = np.asarray(mel_if, dtype=dtype) * dt(1.0) + dt(0.0)          # :108
    mel_mag = np.exp(log_mel)                                             # :110
    mel_phase = np.cumsum(mel_if * pi, axis=-2, dtype=dtype)              # :111
    mel = linear_to_mel_weight_matrix(nbins, nbins, sample_rate, 0.0, sample_rate / 2.0, dtype)
    mel_inv = pinv(mel) if mel_inverse is None else np.asarray(mel_inverse, dtype=dtype)   # :122
    mag = (mel_mag @ mel_inv).astype(dtype)                               # :123
    phase = (mel_phase @ mel_inv).astype(dtype)                           # :125
    s = mag * (np.cos(phase) + 1j * np.sin(phase))                        # :128
    s = np.pad(s, [[0, 0], [0, 0], [1, 0]])                               # :131
    wave = inverse_stft(s, frame_length, frame_step, dtype)               # :132-144
    return wave[:, num_samples - waveform_length:]                        # :147


def cross_correlation(x, y):
    """spectral_ops.py:152-174 with padding="VALID", normalize=True on equal-length
    signals: the single zero-lag product of the l2-normalised signals."""
    x = np.asarray(x, np.float64)
    y = np.asarray(y, np.float64)
    xn = x / np.maximum(np.sqrt((x * x).sum(-1, keepdims=True)), 1e-12)
    yn = y / np.maximum(np.sqrt((y * y).sum(-1, keepdims=True)), 1e-12)
    return (xn * yn).sum(-1)
