"""numpy float64 direct-definition restatement of reference ops.py:149-348.

Test infrastructure (see oracle/__init__.py).  Tensors are NCHW like the
reference.  Every function is written from the mathematical definition of the TF
op it replaces (explicit index arithmetic), NOT from a library conv, so that it is
an independent check of oracle/torch_ref.py.
"""
import numpy as np


def weight_scale(shape, variance_scale):
    """ops.py:154  stddev = sqrt(variance_scale / prod(shape[:-1]))."""
    return float(np.sqrt(variance_scale / np.prod(shape[:-1])))


def dense(x, w, b, variance_scale=2.0):
    """ops.py:183-201  x @ (w*scale) + b ; w is [in, out]."""
    return x @ (w * weight_scale(w.shape, variance_scale)) + b


def embedding(labels, w, variance_scale=2.0):
    """ops.py:204-218  embedding_lookup(w*scale, argmax(labels, 1))."""
    return (w * weight_scale(w.shape, variance_scale))[np.argmax(labels, axis=1)]


def same_pads(size, k, s):
    """TF SAME padding: out = ceil(size/s); total = max((out-1)*s + k - size, 0);
    before = total // 2, after = total - before (the odd pad goes at the END)."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w, b, strides=(1, 1), variance_scale=2.0):
    """ops.py:221-247  tf.nn.conv2d NCHW / HWIO, SAME, cross-correlation."""
    n, ci, h, wd = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    ws = w * weight_scale(w.shape, variance_scale)
    oh, pt, pb = same_pads(h, kh, strides[0])
    ow, pl, pr = same_pads(wd, kw, strides[1])
    xp = np.pad(x, ((0, 0), (0, 0), (pt, pb), (pl, pr)))
    y = np.zeros((n, co, oh, ow), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            patch = xp[:, :, dy:dy + (oh - 1) * strides[0] + 1:strides[0],
                       dx:dx + (ow - 1) * strides[1] + 1:strides[1]]
            y += np.einsum("nchw,co->nohw", patch, ws[dy, dx])
    return y + b[None, :, None, None]


def conv2d_transpose(x, w, b, strides=(2, 2), variance_scale=2.0):
    """ops.py:250-280.  tf.nn.conv2d_transpose(SAME, output = in*strides) is the
    gradient w.r.t. the input of the SAME strided conv:  out[s*i + k - pad_before]
    += in[i] * var[k, ci, co] (var is the STORED [kh,kw,Cin,Cout] variable; the
    reference's transpose at ops.py:266 only re-labels it into TF's
    [kh,kw,out,in] filter convention), positions outside [0, s*H) dropped.
    fan-in for the scale is prod(stored_shape[:-1]) = kh*kw*Cin (ops.py:260)."""
    n, ci, h, wd = x.shape
    kh, kw, ci2, co = w.shape
    assert ci == ci2
    ws = w * weight_scale(w.shape, variance_scale)
    sy, sx = strides
    oh, ow = h * sy, wd * sx
    _, pt, _ = same_pads(oh, kh, sy)
    _, pl, _ = same_pads(ow, kw, sx)
    full = np.zeros((n, co, (h - 1) * sy + kh, (wd - 1) * sx + kw), dtype=x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            full[:, :, dy:dy + (h - 1) * sy + 1:sy, dx:dx + (wd - 1) * sx + 1:sx] += \
                np.einsum("nchw,co->nohw", x, ws[dy, dx])
    y = full[:, :, pt:pt + oh, pl:pl + ow]
    return y + b[None, :, None, None]


def upscale2d(x, factors):
    """ops.py:283-291 nearest-neighbour integer upscale (reshape/tile)."""
    fy, fx = factors
    return np.repeat(np.repeat(x, fy, axis=2), fx, axis=3)


def downscale2d(x, factors):
    """ops.py:294-305 avg_pool ksize=strides=factors (exact division => no pad)."""
    fy, fx = factors
    n, c, h, w = x.shape
    return x.reshape(n, c, h // fy, fy, w // fx, fx).mean(axis=(3, 5))


def pixel_normalization(x, epsilon=1e-12):
    """ops.py:330-333."""
    return x / np.sqrt(np.mean(np.square(x), axis=1, keepdims=True) + epsilon)


def batch_stddev(x, groups=4, epsilon=1e-12):
    """ops.py:336-348 (reshape [groups, -1, C, H, W]: members of a group are
    samples i, i+B/4, i+2B/4, i+3B/4)."""
    shape = x.shape
    y = x.reshape(groups, -1, *shape[1:])
    y = y - y.mean(axis=0, keepdims=True)
    y = np.square(y).mean(axis=0)
    y = np.sqrt(y + epsilon)
    y = y.mean(axis=(1, 2, 3), keepdims=True)
    return np.tile(y, (groups, 1, *shape[2:]))


def leaky_relu(x, alpha=0.2):
    """tf.nn.leaky_relu default alpha=0.2 (networks.py:55,...)."""
    return np.where(x >= 0, x, alpha * x)


def lerp(a, b, t):
    """networks.py:10-11."""
    return t * a + (1.0 - t) * b
