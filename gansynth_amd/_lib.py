"""ctypes binding of libgansynth_hip.so (C ABI declared in include/gansynth_hip.h).

Fails loudly: if the shared library has not been built (`python -c 'import __graft_entry__ as g;
g.build()'` or gansynth_amd/csrc/build.sh) loading raises -- there is no fallback path.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_void_p, POINTER

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgansynth_hip.so")

GS_F32, GS_BF16 = 0, 1
ACT_NONE, ACT_LRELU, ACT_TANH = 0, 1, 2
ACT_LRELU_BITS, ACT_WRITE_BITS = 5, 16   # include/gansynth_hip.h: 1-bit leaky-relu masks
PREP_CONV_FWD, PREP_CONV_BWD_DATA, PREP_CONVT_FWD, PREP_CONVT_BWD_DATA = 0, 1, 2, 3
CONV_FWD, CONV_BWD_DATA, CONV_BWD_WEIGHT = 0, 1, 2

P, I, F, L, Z = c_void_p, c_int, c_float, c_int64, c_size_t

# name -> (restype, argtypes); every symbol include/gansynth_hip.h declares
SIGNATURES = {
    "gs_last_error": (c_char_p, []),
    "gs_version": (I, []),
    "gs_init": (I, []),
    "gs_streams_create": (I, [I, POINTER(P)]),
    "gs_streams_destroy": (I, [I, POINTER(P)]),
    "gs_prof_enable": (I, [I]),
    "gs_prof_collect": (I, [POINTER(c_int), POINTER(c_double), POINTER(c_double)]),
    "gs_prof_roofline": (I, [ctypes.c_double, ctypes.c_double, P, P, P]),
    "gs_prof_records": (I, [I, P, P, P, P, P]),
    "gs_comm_available": (I, []),
    "gs_comm_unique_id": (I, [P]),
    "gs_comm_init": (I, [POINTER(P), I, I, P]),
    "gs_comm_destroy": (I, [P]),
    "gs_comm_count": (I, [P, POINTER(I)]),
    "gs_allreduce_sum_f32": (I, [P, P, ctypes.c_int64, P]),
    "gs_comm_set_marker_us": (I, [P, ctypes.c_double]),
    "gs_broadcast_f32": (I, [P, P, ctypes.c_int64, I, P]),
    "gs_conv2d_workspace_bytes": (Z, [I, I, I, I, I, I, I, I, I]),
    "gs_conv2d_fwd": (I, [P, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_fwd_bias_act": (I, [P, P, P, P, I, I, I, I, I, I, I, F, I, I, I, P, Z, P]),
    "gs_conv2d_bwd_data": (I, [P, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_fwd_bias_act_norm": (I, [P, P, P, P, P, I, I, I, I, I, I, I, F, I, F, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_fwd_bias_act_norm": (I, [P, P, P, P, P, I, I, I, I, I, F, I, F, I, I, P, Z, P]),
    "gs_conv2d_fwd_mask": (I, [P, P, P, I, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_data_mask": (I, [P, P, P, I, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_weight": (I, [P, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_weight_bias": (I, [P, P, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_weight_bias_partial": (I, [P, P, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P, P]),
    "gs_conv2d_transpose_s2_bwd_weight_partial": (I, [P, P, P, I, I, I, I, I, F, I, I, P, Z, P, P]),
    "gs_wgrad_reduce_batch": (I, [P, I, P]),
    "gs_conv2d_bwd_weight_bias_multi": (I, [P, P, P, I, ctypes.c_uint, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P, P]),
    "gs_conv2d_transpose_s2_bwd_weight_multi": (I, [P, P, P, I, P, I, I, I, I, I, F, I, I, P, Z, P, P]),
    "gs_conv_wgrad_jobs_workspace_bytes": (Z, [P, I]),
    "gs_conv_wgrad_jobs": (I, [P, I, P, Z, P]),
    "gs_wgrad_cu_cap": (I, [I]),
    "gs_conv2d_transpose_s2_workspace_bytes": (Z, [I, I, I, I, I, I, I]),
    "gs_conv2d_transpose_s2_fwd": (I, [P, P, P, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_fwd_bias_act": (I, [P, P, P, P, I, I, I, I, I, F, I, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_bwd_data": (I, [P, P, P, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_data_pnbwd_is_fused": (I, [I, I, I, I, I, I, I, I, I]),
    "gs_conv2d_fwd_pnbwdbwd_is_fused": (I, [I, I, I, I, I, I, I, I, I]),
    "gs_units_bias_act_to_nhwc": (I, [P, P, P, P, I, I, I, I, I, P]),
    "gs_nhwc_act_bwd_to_units": (I, [P, P, P, I, I, I, I, I, P]),
    "gs_conv2d_fwd_pnbwdbwd": (I, [P, P, P, P, I, F, P, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_fwd_pnbwdbwd": (I, [P, P, P, P, I, F, P, P, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_bwd_data_pnbwd": (I, [P, P, P, P, I, F, P, I, I, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_bwd_data_pnbwd": (I, [P, P, P, P, I, F, P, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_conv2d_transpose_s2_bwd_weight": (I, [P, P, P, I, I, I, I, I, F, I, I, P, Z, P]),
    "gs_dense_fwd_workspace_bytes": (Z, [I, I, I]),
    "gs_dense_fwd": (I, [P, P, P, I, I, I, F, I, P, Z, P]),
    "gs_dense_fwd_bias_act": (I, [P, P, P, P, I, I, I, F, I, I, P, Z, P]),
    "gs_dense_fwd_bias_act_nhwc": (I, [P, P, P, P, I, I, I, I, F, I, I, P, Z, P]),
    "gs_dense_bwd_data": (I, [P, P, P, I, I, I, F, I, P]),
    "gs_dense_bwd_weight": (I, [P, P, P, I, I, I, F, I, I, P]),
    "gs_dense_fwd_nhwc": (I, [P, P, P, I, I, I, I, F, I, P, Z, P]),
    "gs_dense_bwd_data_nhwc": (I, [P, P, P, I, I, I, I, F, I, P]),
    "gs_dense_bwd_weight_nhwc": (I, [P, P, P, I, I, I, I, F, I, I, P]),
    "gs_embedding_fwd": (I, [P, P, P, I, I, I, F, I, P]),
    "gs_embedding_onehot_fwd": (I, [P, P, P, P, I, I, I, F, I, P]),
    "gs_embedding_bwd": (I, [P, P, P, I, I, I, F, I, P]),
    "gs_bias_act_fwd": (I, [P, P, P, L, I, I, I, P]),
    "gs_act_bwd": (I, [P, P, P, L, I, I, P]),
    "gs_act_bwd_bias": (I, [P, P, P, P, L, I, I, I, I, P, Z, P]),
    "gs_tanh_bwd_bwd": (I, [P, P, P, P, L, I, P]),
    "gs_channel_sum_workspace_bytes": (Z, [L, I]),
    "gs_channel_sum": (I, [P, P, L, I, I, I, P, Z, P]),
    "gs_bias_partial_rows": (I, [I, L, I, I]),
    "gs_channel_fold_batch_workspace_bytes": (Z, [P, I]),
    "gs_channel_fold_batch": (I, [P, I, P, Z, P]),
    "gs_pixel_norm_fwd": (I, [P, P, L, I, F, I, P]),
    "gs_pixel_norm_bwd": (I, [P, P, P, L, I, F, I, P]),
    "gs_pixel_norm_bwd_fused": (I, [P, P, P, P, L, I, F, I, I, I, P]),
    "gs_pixel_norm_bwd_bias_workspace_bytes": (Z, [L, I, I]),
    "gs_pixel_norm_bwd_fused_bias": (I, [P, P, P, P, P, L, I, F, I, I, I, I, P, Z, P]),
    "gs_pixel_norm_bwd_bwd_fused": (I, [P, P, P, P, P, L, I, F, I, I, P]),
    "gs_pixel_norm_bwd_bwd": (I, [P, P, P, P, L, I, F, I, P]),
    "gs_upscale2d": (I, [P, P, I, I, I, I, I, I, F, I, P]),
    "gs_blocksum2d": (I, [P, P, I, I, I, I, I, I, F, I, P]),
    "gs_batch_stddev_fwd": (I, [P, P, I, I, I, F, I, P]),
    "gs_batch_stddev_bwd": (I, [P, P, P, P, I, I, I, F, I, P]),
    "gs_batch_stddev_bwd_bwd": (I, [P, P, P, P, P, I, I, I, F, I, P]),
    "gs_axpby": (I, [P, P, P, L, F, F, I, P]),
    "gs_axpby_dev": (I, [P, P, P, L, P, I, I, I, P]),
    "gs_sumsq_rows_workspace_bytes": (Z, [I]),
    "gs_sumsq_rows": (I, [P, P, I, L, I, P, Z, P]),
    "gs_row_scale": (I, [P, P, F, P, I, L, I, P]),
    "gs_weight_prep_batch": (I, [P, I, P]),
    "gs_gan_d_loss": (I, [P, P, P, P, F, I, I, P, P, P, P, I, P]),
    "gs_gan_g_loss": (I, [P, P, P, F, F, I, I, P, P, P, I, P]),
    "gs_adam_tf_step": (I, [P, P, P, P, L, F, F, F, F, F, P]),
    "gs_adam_tf_step_zero_grad": (I, [P, P, P, P, L, F, F, F, F, F, P]),
    "gs_adam_tf_step_dev": (I, [P, P, P, P, L, P, F, F, F, F, I, P]),
    "gs_pack_act_bits": (I, [P, L, I, I, P]),
    "gs_spectral_plan_create": (I, [POINTER(c_void_p), I, I, I, P, P]),
    "gs_spectral_plan_destroy": (I, [P]),
    "gs_stft_fwd": (I, [P, P, I, I, I, P, P, P]),
    "gs_mel_project": (I, [P, P, P, L, P]),
    "gs_if_unwrap": (I, [P, P, P, I, P]),
    "gs_stft_mel_if_fwd": (I, [P, P, I, I, I, P, I, P, Z, P]),
    "gs_stft_mel_if_workspace_bytes": (Z, [P, I]),
    "gs_mel_if_to_waveform": (I, [P, P, I, I, I, P, I, P, Z, P]),
    "gs_mel_if_to_waveform_workspace_bytes": (Z, [P, I]),
}

WGRAD_MAX_SOURCES = 4   # GS_WGRAD_MAX_SOURCES


class GsWgradReduce(ctypes.Structure):
    """include/gansynth_hip.h: one pending slice reduction of a weight gradient."""
    _fields_ = [("partials", c_void_p), ("gw", c_void_p), ("gb", c_void_p), ("nslices", ctypes.c_int32), ("taps", ctypes.c_int32),
                ("ic", ctypes.c_int32), ("oc", ctypes.c_int32), ("alpha", c_float), ("transpose", ctypes.c_int32),
                ("accumulate", ctypes.c_int32), ("ic_ld", ctypes.c_int32)]


SUM_PARTIALS = 2   # GS_SUM_PARTIALS
BIAS_FROM_CHANNEL_SUM, BIAS_FROM_ACT_BWD, BIAS_FROM_PIXEL_NORM_BWD = 0, 1, 2


class GsFoldJob(ctypes.Structure):
    """include/gansynth_hip.h: one pending fold of bias-gradient partial rows for gs_channel_fold_batch."""
    _fields_ = [("part", c_void_p), ("out", c_void_p), ("nparts", ctypes.c_int32), ("c", ctypes.c_int32), ("accumulate", ctypes.c_int32),
                ("reserved", ctypes.c_int32)]


class GsWgradJob(ctypes.Structure):
    """include/gansynth_hip.h: one layer's weight gradient (up to WGRAD_MAX_SOURCES (x, gy) pairs) for gs_conv_wgrad_jobs."""
    _fields_ = [("x", c_void_p * WGRAD_MAX_SOURCES), ("gy", c_void_p * WGRAD_MAX_SOURCES), ("n", ctypes.c_int32 * WGRAD_MAX_SOURCES),
                ("nsrc", ctypes.c_int32), ("bias_mask", ctypes.c_uint32), ("gw", c_void_p), ("gb", c_void_p),
                ("h", ctypes.c_int32), ("w", ctypes.c_int32), ("ci", ctypes.c_int32), ("co", ctypes.c_int32), ("ksize", ctypes.c_int32),
                ("stride", ctypes.c_int32), ("transposed", ctypes.c_int32), ("alpha", c_float), ("accumulate", ctypes.c_int32),
                ("dtype", ctypes.c_int32), ("gw_ci_stride", ctypes.c_int32)]


_lib = None


class GansynthHipError(RuntimeError):
    pass


def load():
    """Load the library once and attach prototypes.  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise GansynthHipError(
            f"{LIB_PATH} not found: build it with gansynth_amd/csrc/build.sh (or __graft_entry__.build()). "
            "gansynth_amd has no CPU/torch fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        msg = load().gs_last_error()
        raise GansynthHipError(f"{what} failed ({code}): {msg.decode() if msg else ''}")
