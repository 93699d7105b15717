"""waveform <-> (log-mel magnitude, instantaneous frequency) on HIP kernels.

Call surface of reference spectral_ops.py:45-149: `convert_to_spectrogram(waveforms,
waveform_length, sample_rate, spectrogram_shape, overlap)` and `convert_to_waveform(log_mel, IF,
...same...)`.  The constant tables (periodic Hann, twiddles, HTK mel matrix in float32 exactly as
tf.signal.linear_to_mel_weight_matrix builds it, its tfp.math.pinv) are built once on the host and
live in an immutable device plan; everything per batch is HIP (csrc/spectral.hip).

No gradient ever flows through these (models.py:47 differentiates w.r.t. real_images), so they
are plain functions, not autograd Functions.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import kernels


# ----------------------------------------------------------------------- host-side constants
def linear_to_mel_weight_matrix(num_mel_bins, num_spectrogram_bins, sample_rate, lower_edge_hertz, upper_edge_hertz):
    """tf.signal.linear_to_mel_weight_matrix as of TF 1.13 (called at spectral_ops.py:76-82),
    float32 arithmetic: HTK mel = 1127 ln(1 + f/700), DC bin excluded then re-added as a zero row."""
    f32 = np.float32

    def mel(f):
        return (f32(1127.0) * np.log(f32(1.0) + np.asarray(f, f32) / f32(700.0))).astype(f32)

    linear = np.linspace(f32(0.0), f32(sample_rate / 2.0), num_spectrogram_bins, dtype=f32)[1:]
    spec_mel = mel(linear)[:, None]
    edges = np.linspace(mel(lower_edge_hertz), mel(upper_edge_hertz), num_mel_bins + 2, dtype=f32)
    lower, center, upper = edges[None, :-2], edges[None, 1:-1], edges[None, 2:]
    weights = np.maximum(f32(0.0), np.minimum((spec_mel - lower) / (center - lower), (upper - spec_mel) / (upper - center)))
    return np.pad(weights, [[1, 0], [0, 0]]).astype(f32)


def pinv(matrix):
    """tfp.math.pinv (spectral_ops.py:122): singular values <= 10*max(shape)*eps*s_max are dropped."""
    rcond = 10.0 * max(matrix.shape) * np.finfo(matrix.dtype).eps
    u, s, vt = np.linalg.svd(matrix.astype(np.float64), full_matrices=False)
    keep = s > rcond * s.max()
    s_inv = np.where(keep, 1.0 / np.where(keep, s, 1.0), 0.0)
    return ((vt.T * s_inv) @ u.T).astype(matrix.dtype)


def _geometry(waveform_length, spectrogram_shape, overlap):
    """spectral_ops.py:50-53 / 102-105."""
    time_steps, num_freq_bins = spectrogram_shape
    frame_length = num_freq_bins * 2
    frame_step = int((1.0 - overlap) * frame_length)
    num_samples = frame_step * (time_steps - 1) + frame_length
    return time_steps, num_freq_bins, frame_length, frame_step, num_samples - waveform_length


class _Plan(object):
    def __init__(self, sample_rate, time_steps, num_freq_bins, frame_length, frame_step, with_inverse):
        self.lib = kernels.get().lib
        mel = linear_to_mel_weight_matrix(num_freq_bins, num_freq_bins, sample_rate, 0.0, sample_rate / 2.0)
        self.mel = np.ascontiguousarray(mel)
        self.mel_pinv = np.ascontiguousarray(pinv(mel)) if with_inverse else None
        self.handle = ctypes.c_void_p()
        _lib.check(self.lib.gs_spectral_plan_create(
            ctypes.byref(self.handle), frame_length, frame_step, time_steps,
            self.mel.ctypes.data_as(ctypes.c_void_p),
            self.mel_pinv.ctypes.data_as(ctypes.c_void_p) if with_inverse else None), "gs_spectral_plan_create")
        self.time_steps, self.nbins, self.frame_length = time_steps, num_freq_bins, frame_length

    def __del__(self):
        try:
            self.lib.gs_spectral_plan_destroy(self.handle)
        except Exception:
            pass


_PLANS = {}


def _plan(sample_rate, spectrogram_shape, overlap, waveform_length, with_inverse=False):
    time_steps, nbins, frame_length, frame_step, _ = _geometry(waveform_length, spectrogram_shape, overlap)
    key = (torch.cuda.current_device(), sample_rate, time_steps, nbins, frame_length, frame_step, with_inverse)
    if key not in _PLANS:
        _PLANS[key] = _Plan(sample_rate, time_steps, nbins, frame_length, frame_step, with_inverse)
    return _PLANS[key]


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dtype_id(dtype):
    return _lib.GS_F32 if dtype == torch.float32 else _lib.GS_BF16


# ------------------------------------------------------------------------------ forward
def convert_to_images(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap, dtype=torch.float32):
    """Fused waveform -> stack([log_mel, IF], axis=1) (models.py:27-28): returns the [B,2,T,F] image
    tensor (channels-last storage, i.e. the kernel's [b][T][F][2] buffer) in one pass."""
    plan = _plan(sample_rate, spectrogram_shape, overlap, waveform_length)
    _, _, _, _, front_pad = _geometry(waveform_length, spectrogram_shape, overlap)
    wave = waveforms.float().contiguous()
    batch = wave.shape[0]
    images = torch.empty((batch, plan.time_steps, plan.nbins, 2), dtype=dtype, device=wave.device)
    nbytes = plan.lib.gs_stft_mel_if_workspace_bytes(plan.handle, batch)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=wave.device)
    _lib.check(plan.lib.gs_stft_mel_if_fwd(plan.handle, wave.data_ptr(), batch, wave.shape[1], front_pad, images.data_ptr(),
                                           _dtype_id(dtype), ws.data_ptr(), ws.numel(), _stream()), "gs_stft_mel_if_fwd")
    return images.permute(0, 3, 1, 2)


def convert_to_spectrogram(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap):
    """spectral_ops.py:45-94 -> (log_mel_magnitude_spectrograms, mel_instantaneous_frequencies), each [B,T,F]."""
    images = convert_to_images(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap)
    return images[:, 0], images[:, 1]


# stage-wise entry points (parity tests against the oracle's intermediates)
def stft_magnitude_phase(waveforms, waveform_length, sample_rate, spectrogram_shape, overlap):
    """spectral_ops.py:57-72: front pad, STFT, drop DC, abs / angle -> ([B,T,F], [B,T,F])."""
    plan = _plan(sample_rate, spectrogram_shape, overlap, waveform_length)
    _, _, _, _, front_pad = _geometry(waveform_length, spectrogram_shape, overlap)
    wave = waveforms.float().contiguous()
    batch = wave.shape[0]
    mag = torch.empty((batch, plan.time_steps, plan.nbins), dtype=torch.float32, device=wave.device)
    phase = torch.empty_like(mag)
    _lib.check(plan.lib.gs_stft_fwd(plan.handle, wave.data_ptr(), batch, wave.shape[1], front_pad, mag.data_ptr(), phase.data_ptr(), _stream()),
               "gs_stft_fwd")
    return mag, phase


def mel_project(spectrograms, waveform_length, sample_rate, spectrogram_shape, overlap):
    """spectral_ops.py:83-86: tensordot with the (sparse) mel matrix."""
    plan = _plan(sample_rate, spectrogram_shape, overlap, waveform_length)
    x = spectrograms.float().contiguous()
    out = torch.empty_like(x)
    _lib.check(plan.lib.gs_mel_project(plan.handle, x.data_ptr(), out.data_ptr(), x.numel() // plan.nbins, _stream()), "gs_mel_project")
    return out


def instantaneous_frequency(mel_phases, waveform_length, sample_rate, spectrogram_shape, overlap):
    """spectral_ops.py:36-44 along the time axis (-2)."""
    plan = _plan(sample_rate, spectrogram_shape, overlap, waveform_length)
    x = mel_phases.float().contiguous()
    out = torch.empty_like(x)
    _lib.check(plan.lib.gs_if_unwrap(plan.handle, x.data_ptr(), out.data_ptr(), x.shape[0], _stream()), "gs_if_unwrap")
    return out


# ------------------------------------------------------------------------------ inverse
def convert_images_to_waveform(images, waveform_length, sample_rate, spectrogram_shape, overlap):
    """[B,2,T,F] images -> waveforms [B, waveform_length] (spectral_ops.py:97-149)."""
    plan = _plan(sample_rate, spectrogram_shape, overlap, waveform_length, with_inverse=True)
    _, _, _, _, front_pad = _geometry(waveform_length, spectrogram_shape, overlap)
    images = images.contiguous(memory_format=torch.channels_last)
    batch = images.shape[0]
    wave = torch.empty((batch, waveform_length), dtype=torch.float32, device=images.device)
    nbytes = plan.lib.gs_mel_if_to_waveform_workspace_bytes(plan.handle, batch)
    ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=images.device)
    _lib.check(plan.lib.gs_mel_if_to_waveform(plan.handle, images.data_ptr(), batch, waveform_length, front_pad, wave.data_ptr(),
                                              _dtype_id(images.dtype), ws.data_ptr(), ws.numel(), _stream()), "gs_mel_if_to_waveform")
    return wave


def convert_to_waveform(log_mel_magnitude_spectrograms, mel_instantaneous_frequencies, waveform_length, sample_rate,
                        spectrogram_shape, overlap):
    """spectral_ops.py:97-149."""
    images = torch.stack([log_mel_magnitude_spectrograms, mel_instantaneous_frequencies], dim=1)
    return convert_images_to_waveform(images, waveform_length, sample_rate, spectrogram_shape, overlap)
