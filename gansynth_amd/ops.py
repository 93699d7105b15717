"""The reference's ops.py call surface on HIP kernels.

Function names, keyword arguments and defaults follow reference ops.py:149-348 (dense,
embedding, conv2d, conv2d_transpose, upscale2d, downscale2d, pixel_normalization,
batch_stddev, get_weight, get_bias); tensors are eager torch tensors of logical shape NCHW
(physically channels-last) instead of graph tensors, parameters live in variables.VariableStore
under the reference's scope names.  Two documented extensions keep the HBM traffic down:
  * get_weight returns (variable, runtime_scale): the equalized-LR multiply of ops.py:154-160 is
    folded into the consuming kernel as `alpha` instead of materialising weight*scale;
  * conv2d / conv2d_transpose / dense take `activation=None|"leaky_relu"|"tanh"`, fusing the
    tf.nn.leaky_relu / tf.nn.tanh the reference applies right after them (networks.py:55,66,80,
    91,106,184,194,216,227,241) into the bias pass.  With the default None they behave exactly
    like the reference functions.
"""
import numpy as np
import torch

from . import functional as F
from . import variables
from ._lib import ACT_LRELU, ACT_NONE, ACT_TANH

_ACT = {None: ACT_NONE, "leaky_relu": ACT_LRELU, "tanh": ACT_TANH}


def _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization):
    """ops.py:153-154,168-171: the two weight normalisers every layer function of the reference accepts.  They are accepted here in the
    same positions so that a reference call site runs unchanged; the GANSynth graph never sets them (networks.py passes neither), the
    normalisers themselves (ops.py:5-146) are not on this path, and asking for one is refused rather than ignored."""
    if apply_weight_standardization:
        raise NotImplementedError("apply_weight_standardization=True (ops.py:5-25) is not on the GANSynth hot path: networks.py never sets it")
    if apply_spectral_normalization:
        raise NotImplementedError("apply_spectral_normalization=True (ops.py:28-146) is not on the GANSynth hot path: networks.py never sets it")


def get_weight(shape, variance_scale=2.0, scale_weight=False, apply_weight_standardization=False, apply_spectral_normalization=False):
    """ops.py:149-171.  Returns (variable, alpha): alpha = sqrt(variance_scale / prod(shape[:-1]))
    when scale_weight (variable ~ truncN(0,1)), else 1 (variable ~ truncN(0, stddev))."""
    _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization)
    stddev = float(np.sqrt(variance_scale / np.prod(shape[:-1])))
    store = variables.default_store()
    if scale_weight:
        return store.get_variable("weight", shape, variables.truncated_normal(0.0, 1.0)), stddev
    return store.get_variable("weight", shape, variables.truncated_normal(0.0, stddev)), 1.0


def get_bias(shape):
    """ops.py:174-180."""
    return variables.default_store().get_variable("bias", shape, variables.zeros())


def dense(inputs, units, use_bias=True, variance_scale=2.0, scale_weight=False, apply_weight_standardization=False,
          apply_spectral_normalization=False, activation=None):
    """ops.py:183-201."""
    _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization)
    if inputs.dim() == 4:   # tf.layers.flatten of NCHW (networks.py:185) folded into the layer: no channel-major copy
        weight, alpha = get_weight([inputs.shape[1] * inputs.shape[2] * inputs.shape[3], units], variance_scale, scale_weight)
    else:
        weight, alpha = get_weight([inputs.shape[1], units], variance_scale, scale_weight)
    bias = get_bias([units]) if use_bias else None
    if bias is not None or activation is not None:   # dense -> bias_add -> activation as one node (bias / activation where the forward writes)
        return F.dense_bias_act(inputs, weight, bias, alpha, _ACT[activation])
    return F.dense_of_flattened(inputs, weight, alpha) if inputs.dim() == 4 else F.dense(inputs, weight, alpha)


def dense_reshaped(inputs, channels, resolution, use_bias=True, variance_scale=2.0, scale_weight=False, activation=None):
    """dense (ops.py:183-201) -> tf.reshape to [-1, channels, *resolution] -> activation (networks.py:44-55): the dense layer's units are
    channel-major, the activation comes out channels-last -- bias, activation and the reorder are one pass (functional._UnitsBiasActNHWC)
    where the kernel layer has it, the three separate steps otherwise.  Same variables ("weight" [in, units], "bias" [units])."""
    h, w = int(resolution[0]), int(resolution[1])
    units = int(channels) * h * w
    if not F.units_nhwc_ok():
        x = dense(inputs, units=units, use_bias=use_bias, variance_scale=variance_scale, scale_weight=scale_weight)
        x = x.reshape(-1, int(channels), h, w)
        return F.bias_act(x, None, _ACT[activation]) if activation is not None else x
    weight, alpha = get_weight([inputs.shape[1], units], variance_scale, scale_weight)
    outputs = F.dense(inputs, weight, alpha)
    bias = get_bias([units]) if use_bias else None
    return F.units_bias_act_nhwc(outputs, bias, channels, h, w, _ACT[activation])


def embedding(inputs, units, variance_scale=2.0, scale_weight=False, apply_weight_standardization=False, apply_spectral_normalization=False):
    """ops.py:204-218: row gather by argmax of the (one-hot) inputs."""
    _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization)
    weight, alpha = get_weight([inputs.shape[1], units], variance_scale, scale_weight)
    return F.embedding_onehot(inputs, weight, alpha)


def conv2d(inputs, filters, kernel_size, strides=[1, 1], use_bias=True, variance_scale=2.0, scale_weight=False,
           apply_weight_standardization=False, apply_spectral_normalization=False,
           activation=None, input_activation=None, pixel_norm_epsilon=None, input_normed=False):
    """ops.py:221-247 (NCHW, SAME).  `input_normed` (with pixel_norm_epsilon): the caller's promise that `inputs` is the pixel-normalised
    output of such a fused block and feeds nothing but this conv -- its backward may then run that block's norm / activation backward in
    the epilogue of this conv's data-gradient kernel (functional.py); results are unchanged.  `pixel_norm_epsilon`: also apply pixel_normalization(., epsilon) to the activated result
    (the generator's conv -> leaky_relu -> pixel_normalization, networks.py:80-87) as one autograd node.  `input_activation`: the caller's promise that `inputs` is the output of a conv block
    with that fused activation and feeds nothing but this conv -- the backward then folds the activation derivative into
    this conv's data-gradient kernel (functional.py, "premasked gradients"); results are unchanged.  "Nothing but" includes
    second-order graphs: a tensor whose consumers' backward is differentiated again (pixel norm under the mode-seeking term)
    receives a second gradient and must not be declared."""
    _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization)
    kernel_size, strides = list(kernel_size), list(strides)
    if kernel_size[0] != kernel_size[1] or strides[0] != strides[1]:
        raise ValueError("conv2d: only square kernels / isotropic strides are on the hot path")
    weight, alpha = get_weight([*kernel_size, inputs.shape[1], filters], variance_scale, scale_weight)
    bias = get_bias([filters]) if use_bias else None
    if pixel_norm_epsilon is not None:
        return F.conv2d_bias_act_norm(inputs, weight, bias, kernel_size[0], strides[0], alpha, _ACT[activation], pixel_norm_epsilon, input_normed)
    if bias is not None or activation is not None:
        return F.conv2d_bias_act(inputs, weight, bias, kernel_size[0], strides[0], alpha, _ACT[activation], _ACT[input_activation], input_normed)
    return F.conv2d(inputs, weight, kernel_size[0], strides[0], alpha)


def conv2d_transpose(inputs, filters, kernel_size, strides=[1, 1], use_bias=True, variance_scale=2.0, scale_weight=False,
                     apply_weight_standardization=False, apply_spectral_normalization=False,
                     activation=None, pixel_norm_epsilon=None, input_normed=False):
    """ops.py:250-280 (NCHW, SAME, output = input * strides); 3x3 / stride 2 is the hot-path case."""
    _no_weight_normalizers(apply_weight_standardization, apply_spectral_normalization)
    if list(kernel_size) != [3, 3] or list(strides) != [2, 2]:
        raise ValueError("conv2d_transpose: the hot path is kernel 3x3, strides 2x2 (networks.py:71-79)")
    weight, alpha = get_weight([*kernel_size, inputs.shape[1], filters], variance_scale, scale_weight)
    bias = get_bias([filters]) if use_bias else None
    if pixel_norm_epsilon is not None:
        return F.conv2d_transpose_bias_act_norm(inputs, weight, bias, alpha, _ACT[activation], pixel_norm_epsilon, input_normed)
    if bias is not None or activation is not None:
        return F.conv2d_transpose_bias_act(inputs, weight, bias, alpha, _ACT[activation])
    return F.conv2d_transpose(inputs, weight, alpha)


def upscale2d(inputs, factors=[2, 2]):
    """ops.py:283-291."""
    factors = np.asanyarray(factors)
    if (factors == 1).all():
        return inputs
    return F.upscale(inputs, int(factors[0]), int(factors[1]))


def downscale2d(inputs, factors=[2, 2]):
    """ops.py:294-305."""
    factors = np.asanyarray(factors)
    if (factors == 1).all():
        return inputs
    return F.avg_pool(inputs, int(factors[0]), int(factors[1]))


def pixel_normalization(inputs, epsilon=1.0e-12, input_activation=None):
    """ops.py:330-333 (axis 1, also for the 2-D latent).  `input_activation`: see conv2d."""
    return F.pixel_norm(inputs, epsilon, _ACT[input_activation])


def batch_stddev(inputs, groups=4, epsilon=1.0e-12):
    """ops.py:336-348."""
    if groups != 4:
        raise ValueError("batch_stddev: the reference graph uses groups=4 (networks.py:174)")
    if inputs.shape[0] % groups:
        raise ValueError(f"batch_stddev: batch {inputs.shape[0]} is not a multiple of groups={groups} (ops.py:341)")
    return F.batch_stddev(inputs, epsilon)


def batch_stddev_tap(inputs, groups=4, epsilon=1.0e-12, sub_batches=1):
    """(inputs, batch_stddev(inputs)): ops.py:336-348 for a caller that also feeds `inputs` to another op -- it uses the returned
    alias there, and the backward sums the two gradients into `inputs` in the statistic's own kernel (functional._BatchStddevTap).
    `sub_batches`: `inputs` is that many batches concatenated along axis 0; the statistic is taken within each of them."""
    if groups != 4:
        raise ValueError("batch_stddev: the reference graph uses groups=4 (networks.py:174)")
    if inputs.shape[0] % (groups * sub_batches):
        raise ValueError(f"batch_stddev: batch {inputs.shape[0]} / {sub_batches} is not a multiple of groups={groups} (ops.py:341)")
    if sub_batches == 1 and not hasattr(F, "batch_stddev_tap"):
        return inputs, F.batch_stddev(inputs, epsilon)
    return F.batch_stddev_tap(inputs, epsilon, sub_batches)


def leaky_relu(inputs):
    """tf.nn.leaky_relu (alpha 0.2)."""
    return F.bias_act(inputs, None, ACT_LRELU)


def tanh(inputs):
    return F.bias_act(inputs, None, ACT_TANH)


def lerp(a, b, t):
    """networks.py:10-11.  `t`: a number, or a functional.DeviceLerp whose weight lives in device memory (hipGraph-safe)."""
    if isinstance(t, F.DeviceLerp):
        ca, cb = t.weights()
        return F.axpby(a, b, ca, cb)
    return F.axpby(a, b, float(t), 1.0 - float(t))
