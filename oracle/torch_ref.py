"""torch-CPU restatement of the reference graph (ops.py, networks.py, models.py).

Test infrastructure (see oracle/__init__.py): the checker for the HIP path and the
"port" CPU baseline timed by bench.py.  Runs on CPU tensors only (fp32 or fp64),
NCHW like the reference, parameters injected as a {tf_variable_name: tensor} dict
in the reference's layouts (conv HWIO [kh,kw,Cin,Cout], dense [in,out]).

First- and second-order gradients come from torch.autograd on these library ops
(`create_graph=True`), which is how tf.gradients composes in models.py:47,60.
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------- ops.py
def weight_scale(shape, variance_scale):
    """ops.py:154."""
    return math.sqrt(variance_scale / float(np.prod(shape[:-1])))


def dense(x, w, b, variance_scale=2.0):
    """ops.py:183-201."""
    return x @ (w * weight_scale(w.shape, variance_scale)) + b


def embedding(labels, w, variance_scale=2.0):
    """ops.py:204-218."""
    return (w * weight_scale(w.shape, variance_scale))[labels.argmax(dim=1)]


def conv2d(x, w, b, strides=(1, 1), variance_scale=2.0):
    """ops.py:221-247.  TF SAME: stride 1 pads (k-1)/2 both sides; stride 2 on an even
    input pads 0 before / 1 after (odd pad at the end) -- SURVEY.md N4."""
    ws = (w * weight_scale(w.shape, variance_scale)).permute(3, 2, 0, 1)  # HWIO -> OIHW
    kh, kw = w.shape[0], w.shape[1]
    pads = []
    for size, k, s in ((x.shape[3], kw, strides[1]), (x.shape[2], kh, strides[0])):
        out = -(-size // s)
        total = max((out - 1) * s + k - size, 0)
        pads += [total // 2, total - total // 2]
    y = F.conv2d(F.pad(x, pads), ws, stride=tuple(strides))
    return y + b.view(1, -1, 1, 1)


def conv2d_transpose(x, w, b, strides=(2, 2), variance_scale=2.0):
    """ops.py:250-280.  out[2i+k] += in[i]*var[k,ci,co], cropped at the END to 2H x 2W
    (pad_before of the matching SAME stride-2 conv is 0) -- SURVEY.md N5."""
    ws = (w * weight_scale(w.shape, variance_scale)).permute(2, 3, 0, 1)  # -> [Cin,Cout,kh,kw]
    y = F.conv_transpose2d(x, ws, stride=tuple(strides))
    y = y[:, :, : x.shape[2] * strides[0], : x.shape[3] * strides[1]]
    return y + b.view(1, -1, 1, 1)


def upscale2d(x, factors):
    """ops.py:283-291."""
    fy, fx = int(factors[0]), int(factors[1])
    if fy == 1 and fx == 1:
        return x
    return x.repeat_interleave(fy, dim=2).repeat_interleave(fx, dim=3)


def downscale2d(x, factors):
    """ops.py:294-305."""
    fy, fx = int(factors[0]), int(factors[1])
    if fy == 1 and fx == 1:
        return x
    return F.avg_pool2d(x, kernel_size=(fy, fx), stride=(fy, fx))


def pixel_normalization(x, epsilon=1e-12):
    """ops.py:330-333."""
    return x / torch.sqrt(torch.mean(x * x, dim=1, keepdim=True) + epsilon)


def batch_stddev(x, groups=4, epsilon=1e-12):
    """ops.py:336-348."""
    shape = x.shape
    y = x.reshape(groups, -1, *shape[1:])
    y = y - y.mean(dim=0, keepdim=True)
    y = (y * y).mean(dim=0)
    y = torch.sqrt(y + epsilon)
    y = y.mean(dim=(1, 2, 3), keepdim=True)
    return y.repeat(groups, 1, *shape[2:])


class lrelu_tape(object):
    """Test instrumentation for the one ill-conditioned op of the graph.  leaky_relu is piecewise linear: a pre-activation
    that differs in its last fp32 bits between two implementations can land on either side of 0, which moves that unit's
    gradient 5x.  `lrelu_tape("record")` keeps, per network call (`calls[i] = (name, [x, ...])`, in call order), every
    leaky_relu input; `lrelu_tape("override", masks)` evaluates leaky_relu as where(mask, x, 0.2 x) with the masks of another
    implementation (same nesting), so that the two can be compared on the same linear pieces."""
    active = None

    def __init__(self, mode, masks=None):
        self.mode, self.masks, self.calls = mode, masks, []
        self._cursor = None

    def __enter__(self):
        lrelu_tape.active = self
        return self

    def __exit__(self, *exc):
        lrelu_tape.active = None
        return False

    def begin(self, name):
        self.calls.append((name, []))
        if self.mode == "override":
            want = self.masks[len(self.calls) - 1]
            assert want[0] == name, (want[0], name)
            self._cursor = iter(want[1])


def leaky_relu(x):
    tape = lrelu_tape.active
    if tape is None:
        return F.leaky_relu(x, 0.2)
    if tape.mode == "record":
        tape.calls[-1][1].append(x.detach())
        return F.leaky_relu(x, 0.2)
    mask = next(tape._cursor)
    assert mask.shape == x.shape, (tuple(mask.shape), tuple(x.shape))
    return torch.where(mask, x, x * 0.2)


def lerp(a, b, t):
    """networks.py:10-11."""
    return t * a + (1.0 - t) * b


# ------------------------------------------------------------------------- networks.py
class PGGAN(object):
    """networks.py:14-290 with host-side branch selection (tf.cond executes only the
    taken branch; the predicate is `growing_depth > depth`, strict)."""

    def __init__(self, min_resolution, max_resolution, min_channels, max_channels, growing_level):
        self.min_resolution = np.asanyarray(min_resolution)
        self.max_resolution = np.asanyarray(max_resolution)
        self.min_channels = min_channels
        self.max_channels = max_channels
        self.growing_level = growing_level

        def log2(x):
            return 0 if (x == 1).all() else 1 + log2(x >> 1)

        self.min_depth = log2(self.min_resolution // self.min_resolution)
        self.max_depth = log2(self.max_resolution // self.min_resolution)

    @property
    def growing_depth(self):
        """networks.py:29 (float32 arithmetic in TF)."""
        level = float(self.growing_level() if callable(self.growing_level) else self.growing_level)
        return float(np.log2(np.float32(1.0) + np.float32((1 << (self.max_depth + 1)) - 1) * np.float32(level)))

    def resolution(self, depth):
        return self.min_resolution << depth

    def channels(self, depth):
        return min(self.max_channels, self.min_channels << (self.max_depth - depth))

    # -------------------------------------------------------------- parameter inventory
    def variable_shapes(self, latent_dim=256, num_labels=61):
        """All trainable variables of both networks (every tf.cond branch exists from
        step 0), keyed by the TF scope names of networks.py:40-290."""
        g, d = OrderedDict(), OrderedDict()
        g["generator/weight"] = (num_labels, latent_dim)
        for depth in range(self.min_depth, self.max_depth + 1):
            r = "{}x{}".format(*self.resolution(depth))
            c = self.channels(depth)
            if depth == self.min_depth:
                g[f"generator/conv_block_{r}/dense/weight"] = (2 * latent_dim, c * int(self.resolution(depth).prod()))
                g[f"generator/conv_block_{r}/dense/bias"] = (c * int(self.resolution(depth).prod()),)
            else:
                g[f"generator/conv_block_{r}/upscale_conv/weight"] = (3, 3, self.channels(depth - 1), c)
                g[f"generator/conv_block_{r}/upscale_conv/bias"] = (c,)
            g[f"generator/conv_block_{r}/conv/weight"] = (3, 3, c, c)
            g[f"generator/conv_block_{r}/conv/bias"] = (c,)
            g[f"generator/color_block_{r}/conv/weight"] = (1, 1, c, 2)
            g[f"generator/color_block_{r}/conv/bias"] = (2,)
        for depth in range(self.min_depth, self.max_depth + 1):
            r = "{}x{}".format(*self.resolution(depth))
            c = self.channels(depth)
            d[f"discriminator/color_block_{r}/conv/weight"] = (1, 1, 2, c)
            d[f"discriminator/color_block_{r}/conv/bias"] = (c,)
            if depth == self.min_depth:
                d[f"discriminator/conv_block_{r}/conv/weight"] = (3, 3, c + 1, c)
                d[f"discriminator/conv_block_{r}/conv/bias"] = (c,)
                d[f"discriminator/conv_block_{r}/dense/weight"] = (c * int(self.resolution(depth).prod()), self.channels(depth - 1))
                d[f"discriminator/conv_block_{r}/dense/bias"] = (self.channels(depth - 1),)
                d[f"discriminator/conv_block_{r}/logits/weight"] = (self.channels(depth - 1), num_labels)
                d[f"discriminator/conv_block_{r}/logits/bias"] = (num_labels,)
            else:
                d[f"discriminator/conv_block_{r}/conv/weight"] = (3, 3, c, c)
                d[f"discriminator/conv_block_{r}/conv/bias"] = (c,)
                d[f"discriminator/conv_block_{r}/conv_downscale/weight"] = (3, 3, c, self.channels(depth - 1))
                d[f"discriminator/conv_block_{r}/conv_downscale/bias"] = (self.channels(depth - 1),)
        return g, d

    def init_params(self, seed=0, dtype=torch.float32, bias_std=0.0):
        """ops.py:156-160,174-180: weights truncated_normal(0,1) (re-draw beyond 2 sigma),
        biases zero (bias_std>0 draws N(0,bias_std) biases for tests only)."""
        gen = torch.Generator().manual_seed(seed)
        out = []
        for shapes in self.variable_shapes():
            p = OrderedDict()
            for name, shape in shapes.items():
                if name.endswith("bias"):
                    t = torch.randn(shape, generator=gen, dtype=torch.float64) * bias_std
                else:
                    t = torch.empty(shape, dtype=torch.float64)
                    torch.nn.init.trunc_normal_(t, 0.0, 1.0, -2.0, 2.0, generator=gen)
                p[name] = t.to(dtype)
            out.append(p)
        return out

    # ------------------------------------------------------------------------ generator
    def generator(self, params, latents, labels, name="generator"):
        """networks.py:31-161."""
        P = params
        gd = self.growing_depth
        if lrelu_tape.active is not None:
            lrelu_tape.active.begin("generator")

        def rname(depth):
            return "{}x{}".format(*self.resolution(depth))

        def conv_block(x, depth):
            s = f"{name}/conv_block_{rname(depth)}"
            if depth == self.min_depth:
                x = pixel_normalization(x)
                x = dense(x, P[f"{s}/dense/weight"], P[f"{s}/dense/bias"], 2.0)
                x = x.reshape(-1, self.channels(depth), *self.resolution(depth))
                x = pixel_normalization(leaky_relu(x))
                x = conv2d(x, P[f"{s}/conv/weight"], P[f"{s}/conv/bias"], (1, 1), 2.0)
                return pixel_normalization(leaky_relu(x))
            x = conv2d_transpose(x, P[f"{s}/upscale_conv/weight"], P[f"{s}/upscale_conv/bias"], (2, 2), 2.0)
            x = pixel_normalization(leaky_relu(x))
            x = conv2d(x, P[f"{s}/conv/weight"], P[f"{s}/conv/bias"], (1, 1), 2.0)
            return pixel_normalization(leaky_relu(x))

        def color_block(x, depth):
            s = f"{name}/color_block_{rname(depth)}/conv"
            return torch.tanh(conv2d(x, P[f"{s}/weight"], P[f"{s}/bias"], (1, 1), 1.0))

        def grow(x, depth):
            def high():
                return grow(conv_block(x, depth), depth + 1)

            def middle():
                return upscale2d(color_block(conv_block(x, depth), depth),
                                 self.resolution(self.max_depth) // self.resolution(depth))

            def low():
                return upscale2d(color_block(x, depth - 1),
                                 self.resolution(self.max_depth) // self.resolution(depth - 1))

            if depth == self.min_depth:
                return high() if gd > depth else middle()
            if depth == self.max_depth:
                return middle() if gd > depth else lerp(low(), middle(), depth - gd)
            return high() if gd > depth else lerp(low(), middle(), depth - gd)

        emb = embedding(labels, P[f"{name}/weight"], 1.0)
        return grow(torch.cat([latents, emb], dim=1), self.min_depth)

    # -------------------------------------------------------------------- discriminator
    def discriminator(self, params, images, labels, name="discriminator"):
        """networks.py:163-290."""
        P = params
        gd = self.growing_depth
        if lrelu_tape.active is not None:
            lrelu_tape.active.begin("discriminator")

        def rname(depth):
            return "{}x{}".format(*self.resolution(depth))

        def conv_block(x, depth):
            s = f"{name}/conv_block_{rname(depth)}"
            if depth == self.min_depth:
                x = torch.cat([x, batch_stddev(x)], dim=1)
                x = leaky_relu(conv2d(x, P[f"{s}/conv/weight"], P[f"{s}/conv/bias"], (1, 1), 2.0))
                x = x.reshape(x.shape[0], -1)
                feats = leaky_relu(dense(x, P[f"{s}/dense/weight"], P[f"{s}/dense/bias"], 2.0))
                logits = dense(feats, P[f"{s}/logits/weight"], P[f"{s}/logits/bias"], 1.0)
                return feats, logits
            x = leaky_relu(conv2d(x, P[f"{s}/conv/weight"], P[f"{s}/conv/bias"], (1, 1), 2.0))
            return leaky_relu(conv2d(x, P[f"{s}/conv_downscale/weight"], P[f"{s}/conv_downscale/bias"], (2, 2), 2.0))

        def color_block(x, depth):
            s = f"{name}/color_block_{rname(depth)}/conv"
            return leaky_relu(conv2d(x, P[f"{s}/weight"], P[f"{s}/bias"], (1, 1), 2.0))

        def grow(depth):
            def high():
                return conv_block(grow(depth + 1), depth)

            def middle():
                return conv_block(color_block(downscale2d(
                    images, self.resolution(self.max_depth) // self.resolution(depth)), depth), depth)

            def low():
                return color_block(downscale2d(
                    images, self.resolution(self.max_depth) // self.resolution(depth - 1)), depth - 1)

            if depth == self.min_depth:
                return high() if gd > depth else middle()
            if depth == self.max_depth:
                return middle() if gd > depth else lerp(low(), middle(), depth - gd)
            return high() if gd > depth else lerp(low(), middle(), depth - gd)

        return grow(self.min_depth)


# --------------------------------------------------------------------------- models.py
DEFAULT_HYPER = dict(
    generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99,
    discriminator_learning_rate=8e-4, discriminator_beta1=0.0, discriminator_beta2=0.99,
    mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0,
)


def discriminator_loss(pggan, g_params, d_params, latents, labels, real_images, hyper=DEFAULT_HYPER):
    """models.py:25,33-54,65 -- the subgraph discriminator_train_op evaluates."""
    with torch.no_grad():
        fake_images = pggan.generator(g_params, latents, labels)
    real_images = real_images.detach().requires_grad_(True)
    if hyper.get("fake_gradient_penalty_weight", 0.0):
        fake_images = fake_images.detach().requires_grad_(True)
    _, real_logits = pggan.discriminator(d_params, real_images, labels)
    _, fake_logits = pggan.discriminator(d_params, fake_images, labels)
    real_logits = (real_logits * labels).sum(dim=1)  # gather_nd(where(one_hot))
    fake_logits = (fake_logits * labels).sum(dim=1)
    losses = F.softplus(-real_logits) + F.softplus(fake_logits)
    if hyper["real_gradient_penalty_weight"]:
        (grads,) = torch.autograd.grad(real_logits.sum(), real_images, create_graph=True)
        losses = losses + grads.pow(2).sum(dim=(1, 2, 3)) * hyper["real_gradient_penalty_weight"]
    if hyper.get("fake_gradient_penalty_weight", 0.0):   # models.py:50-54
        (grads,) = torch.autograd.grad(fake_logits.sum(), fake_images, create_graph=True)
        losses = losses + grads.pow(2).sum(dim=(1, 2, 3)) * hyper["fake_gradient_penalty_weight"]
    return losses.mean()


def generator_loss(pggan, g_params, d_params, latents, labels, hyper=DEFAULT_HYPER):
    """models.py:25,34,40,57-64 -- the subgraph generator_train_op evaluates."""
    latents = latents.detach().requires_grad_(True)
    fake_images = pggan.generator(g_params, latents, labels)
    _, fake_logits = pggan.discriminator(d_params, fake_images, labels)
    fake_logits = (fake_logits * labels).sum(dim=1)
    losses = F.softplus(-fake_logits)
    if hyper["mode_seeking_loss_weight"]:
        (lg,) = torch.autograd.grad(fake_images.sum(), latents, create_graph=True)
        losses = losses + hyper["mode_seeking_loss_weight"] / (lg.pow(2).sum(dim=1) + 1.0e-6)
    return losses.mean()


def adam_tf_step(params, grads, m, v, step, lr, beta1, beta2, epsilon=1e-8):
    """tf.train.AdamOptimizer (models.py:67-76): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
    theta -= lr_t * m / (sqrt(v) + eps) -- eps is NOT bias-corrected (differs from
    torch.optim.Adam).  `step` is t (1-based).  Updates in place."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    with torch.no_grad():
        for k in params:
            g = grads[k]
            m[k].mul_(beta1).add_(g, alpha=1.0 - beta1)
            v[k].mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            params[k].sub_(lr_t * m[k] / (v[k].sqrt() + epsilon))


class Trainer(object):
    """One reference iteration = session.run(discriminator_train_op) then
    session.run(generator_train_op) (models.py:189-194), each on its own batch."""

    def __init__(self, pggan, g_params, d_params, hyper=DEFAULT_HYPER):
        self.pggan, self.hyper = pggan, dict(hyper)
        self.g = OrderedDict((k, t.detach().clone().requires_grad_(True)) for k, t in g_params.items())
        self.d = OrderedDict((k, t.detach().clone().requires_grad_(True)) for k, t in d_params.items())
        self.gm = {k: torch.zeros_like(t) for k, t in self.g.items()}
        self.gv = {k: torch.zeros_like(t) for k, t in self.g.items()}
        self.dm = {k: torch.zeros_like(t) for k, t in self.d.items()}
        self.dv = {k: torch.zeros_like(t) for k, t in self.d.items()}
        self.g_t = 0
        self.d_t = 0
        self.global_step = 0

    @staticmethod
    def _grads(loss, params):
        keys = list(params)
        gs = torch.autograd.grad(loss, [params[k] for k in keys], allow_unused=True)
        return {k: (torch.zeros_like(params[k]) if g is None else g) for k, g in zip(keys, gs)}

    def d_step(self, latents, labels, real_images):
        loss = discriminator_loss(self.pggan, self.g, self.d, latents, labels, real_images, self.hyper)
        grads = self._grads(loss, self.d)
        self.d_t += 1
        adam_tf_step(self.d, grads, self.dm, self.dv, self.d_t, self.hyper["discriminator_learning_rate"],
                     self.hyper["discriminator_beta1"], self.hyper["discriminator_beta2"])
        return loss.detach(), grads

    def g_step(self, latents, labels):
        loss = generator_loss(self.pggan, self.g, self.d, latents, labels, self.hyper)
        grads = self._grads(loss, self.g)
        self.g_t += 1
        adam_tf_step(self.g, grads, self.gm, self.gv, self.g_t, self.hyper["generator_learning_rate"],
                     self.hyper["generator_beta1"], self.hyper["generator_beta2"])
        self.global_step += 1  # models.py:84: only the generator op bumps global_step
        return loss.detach(), grads


# ----------------------------------------------------------------- synthetic inputs (8d)
def synthetic_batch(batch, rank=0, dtype=torch.float32, image_shape=(2, 128, 1024), latent_dim=256, num_labels=61):
    """latents N(0,1) seed 1000+rank; one-hot labels seed 2000+rank;
    real images ch0 clip(N(-0.2,0.6^2)), ch1 clip(N(0,0.4^2)) seed 3000+rank."""
    g = torch.Generator().manual_seed(1000 + rank)
    latents = torch.randn(batch, latent_dim, generator=g, dtype=torch.float64).to(dtype)
    g = torch.Generator().manual_seed(2000 + rank)
    idx = torch.randint(0, num_labels, (batch,), generator=g)
    labels = F.one_hot(idx, num_labels).to(dtype)
    g = torch.Generator().manual_seed(3000 + rank)
    c, h, w = image_shape
    real = torch.randn(batch, c, h, w, generator=g, dtype=torch.float64)
    real[:, 0] = (real[:, 0] * 0.6 - 0.2)
    real[:, 1] = (real[:, 1] * 0.4)
    real = real.clamp_(-1.0, 1.0).to(dtype)
    return latents, labels, real
