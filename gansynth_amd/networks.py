"""Progressive-GAN generator / discriminator on the HIP ops (reference networks.py:14-290).

Same constructor and call surface as the reference's PGGAN: `generator(latents, labels, name,
reuse)` -> images [B,2,128,1024]; `discriminator(images, labels, name, reuse)` -> (features,
logits).  The reference builds every tf.cond branch into one graph and lets the runtime pick;
here the growing depth is a host number, so each call walks the single active path: a trunk of
conv blocks below the current depth, then either the plain head or the faded (lerp) pair of
heads.  All variables of all depths are still created up front under the reference's scope
names, because the reference's graph owns them from step 0 (zero-gradient Adam updates included).
"""
import numpy as np
import torch

from . import config
from . import functional as F
from . import ops
from . import variables
from .variables import AUTO_REUSE, variable_scope


PIXEL_NORM_EPS = 1.0e-12   # ops.pixel_normalization default (ops.py:330)
_FUSE_NORM = not config.flag("GS_NO_FUSED_NORM")   # A/B switch for measurements


def _ilog2(ratio):
    ratio = np.asanyarray(ratio)
    depth = 0
    while not (ratio == 1).all():
        ratio = ratio >> 1
        depth += 1
    return depth


class PGGAN(object):

    def __init__(self, min_resolution, max_resolution, min_channels, max_channels, growing_level):
        self.min_resolution = np.asanyarray(min_resolution)
        self.max_resolution = np.asanyarray(max_resolution)
        self.min_channels = min_channels
        self.max_channels = max_channels
        self.growing_level = growing_level  # float, or a zero-argument callable (e.g. step / growing_steps)
        self.min_depth = 0
        self.fade_weight = None   # a functional.DeviceLerp: the trainer keeps the fade-in weight in device memory (hipGraph replay)
        self.max_depth = _ilog2(self.max_resolution // self.min_resolution)

    # ------------------------------------------------------------------ schedule (networks.py:24-29)
    @property
    def growing_depth(self):
        level = self.growing_level() if callable(self.growing_level) else self.growing_level
        full = np.float32((1 << (self.max_depth + 1)) - 1)
        return float(np.log2(np.float32(1.0) + full * np.float32(level)))

    def resolution(self, depth):
        return self.min_resolution << depth

    def channels(self, depth):
        return min(self.max_channels, self.min_channels << (self.max_depth - depth))

    def _block_name(self, kind, depth):
        return "{}_block_{}x{}".format(kind, *self.resolution(depth))

    def _head_depth(self, growing_depth):
        """Depth at which the recursion of networks.py:109-152 / 244-287 stops descending, and the
        lerp weight of the low-resolution branch there (None = plain head, no fade)."""
        depth = self.min_depth
        while depth < self.max_depth and growing_depth > depth:
            depth += 1
        if depth == self.min_depth and not growing_depth > depth:
            return depth, None
        if depth == self.max_depth and growing_depth > depth:
            return depth, None
        return depth, depth - growing_depth

    # ================================================================== generator
    def _g_conv_block(self, x, depth, sole_consumer=True):
        """`sole_consumer`: x (the previous block's normalised output) feeds nothing but this block -- false at the fade-in junction, where
        the low-resolution colour block reads it too (lets the backward fuse across the block boundary, ops.conv2d `input_normed`)."""
        c = self.channels(depth)
        with variable_scope(self._block_name("conv", depth)):
            if depth == self.min_depth:
                x = ops.pixel_normalization(x)
                with variable_scope("dense"):   # dense -> reshape -> leaky_relu (networks.py:43-55), then the norm
                    x = ops.dense_reshaped(x, c, self.resolution(depth), use_bias=True, variance_scale=2.0, scale_weight=True, activation="leaky_relu")
                    x = ops.pixel_normalization(x)
            else:
                with variable_scope("upscale_conv"):
                    x = ops.conv2d_transpose(x, filters=c, kernel_size=[3, 3], strides=[2, 2], use_bias=True,
                                             variance_scale=2.0, scale_weight=True, activation="leaky_relu",
                                             pixel_norm_epsilon=PIXEL_NORM_EPS if _FUSE_NORM else None,   # conv -> leaky_relu -> pixel norm: one node
                                             input_normed=sole_consumer)
                    if not _FUSE_NORM:
                        x = ops.pixel_normalization(x)
            with variable_scope("conv"):
                x = ops.conv2d(x, filters=c, kernel_size=[3, 3], use_bias=True, variance_scale=2.0, scale_weight=True,
                               activation="leaky_relu", pixel_norm_epsilon=PIXEL_NORM_EPS if _FUSE_NORM else None,
                               input_normed=depth != self.min_depth)   # (the upscale conv's output feeds this conv only)
                if not _FUSE_NORM:
                    x = ops.pixel_normalization(x)
        return x

    def _g_color_block(self, x, depth, sole_consumer=False):
        """`sole_consumer`: x (a conv block's normalised output) feeds nothing but this colour block (true for the head block's own output)."""
        with variable_scope(self._block_name("color", depth)):
            with variable_scope("conv"):
                return ops.conv2d(x, filters=2, kernel_size=[1, 1], use_bias=True, variance_scale=1.0, scale_weight=True,
                                  activation="tanh", input_normed=sole_consumer and _FUSE_NORM)

    def _g_variables(self, latent_dim, num_labels):
        """Create every generator variable (all depths) in the reference's scopes."""
        ops.get_weight([num_labels, latent_dim], 1.0, True)
        for depth in range(self.min_depth, self.max_depth + 1):
            c = self.channels(depth)
            with variable_scope(self._block_name("conv", depth)):
                if depth == self.min_depth:
                    with variable_scope("dense"):
                        units = c * int(self.resolution(depth).prod())
                        ops.get_weight([2 * latent_dim, units], 2.0, True)
                        ops.get_bias([units])
                else:
                    with variable_scope("upscale_conv"):
                        ops.get_weight([3, 3, self.channels(depth - 1), c], 2.0, True)
                        ops.get_bias([c])
                with variable_scope("conv"):
                    ops.get_weight([3, 3, c, c], 2.0, True)
                    ops.get_bias([c])
            with variable_scope(self._block_name("color", depth)):
                with variable_scope("conv"):
                    ops.get_weight([1, 1, c, 2], 1.0, True)
                    ops.get_bias([2])

    def generator(self, latents, labels, name="generator", reuse=AUTO_REUSE):
        F.tap_begin("generator")
        with variable_scope(name, reuse=reuse):
            self._g_variables(latents.shape[1], labels.shape[1])
            embedded = ops.embedding(labels, units=latents.shape[1], variance_scale=1.0, scale_weight=True)
            x = torch.cat([latents, embedded], dim=1)
            head, fade = self._head_depth(self.growing_depth)
            for depth in range(self.min_depth, head):
                x = self._g_conv_block(x, depth)
            full = self.resolution(self.max_depth)
            middle = ops.upscale2d(self._g_color_block(self._g_conv_block(x, head, sole_consumer=fade is None), head, sole_consumer=True), full // self.resolution(head))
            if fade is None:
                return middle
            low = ops.upscale2d(self._g_color_block(x, head - 1), full // self.resolution(head - 1))
            return ops.lerp(low, middle, fade if self.fade_weight is None else self.fade_weight)

    # ============================================================== discriminator
    def _d_conv_block(self, x, depth, num_labels, fresh_activation=False, sub_batches=1):
        """`fresh_activation`: x is the leaky-relu output of the previous conv and feeds nothing but this block's first conv
        (true along the trunk, false after the fade-in lerp) -- lets the backward fold the activation derivative into the
        data-gradient kernel (ops.conv2d `input_activation`)."""
        c = self.channels(depth)
        with variable_scope(self._block_name("conv", depth)):
            if depth == self.min_depth:
                # networks.py:174-184: conv(concat([x, batch_stddev(x)])).  The 257-channel conv is
                # evaluated as conv(x; w[:,:,:c]) + conv(stddev; w[:,:,c:]) -- same variable, same
                # fan-in scale, no 257-wide tensor (257 is not an MFMA-friendly K).
                x, stddev = ops.batch_stddev_tap(x, sub_batches=sub_batches)   # (x through the tap: one consumer, the two gradients summed in one kernel)
                with variable_scope("conv"):
                    weight, alpha = ops.get_weight([3, 3, c + 1, c], 2.0, True)
                    bias = ops.get_bias([c])
                    y = F.axpby(F.conv2d(x, F.weight_slice(weight, 0, c), 3, 1, alpha),
                                F.conv2d(stddev, F.weight_slice(weight, c, c + 1), 3, 1, alpha), 1.0, 1.0)
                    x = F.bias_act(y, bias, ops._ACT["leaky_relu"])
                with variable_scope("dense"):
                    # tf.layers.flatten of NCHW (channel-major) happens inside ops.dense (4-D input)
                    features = ops.dense(x, units=self.channels(depth - 1), use_bias=True, variance_scale=2.0,
                                         scale_weight=True, activation="leaky_relu")
                with variable_scope("logits"):
                    logits = ops.dense(features, units=num_labels, use_bias=True, variance_scale=1.0, scale_weight=True)
                return features, logits
            with variable_scope("conv"):
                x = ops.conv2d(x, filters=c, kernel_size=[3, 3], use_bias=True, variance_scale=2.0, scale_weight=True,
                               activation="leaky_relu", input_activation="leaky_relu" if fresh_activation else None)
            with variable_scope("conv_downscale"):
                x = ops.conv2d(x, filters=self.channels(depth - 1), kernel_size=[3, 3], strides=[2, 2], use_bias=True,
                               variance_scale=2.0, scale_weight=True, activation="leaky_relu", input_activation="leaky_relu")
            return x

    def _d_color_block(self, x, depth):
        with variable_scope(self._block_name("color", depth)):
            with variable_scope("conv"):
                return ops.conv2d(x, filters=self.channels(depth), kernel_size=[1, 1], use_bias=True, variance_scale=2.0,
                                  scale_weight=True, activation="leaky_relu")

    def _d_variables(self, num_labels):
        for depth in range(self.min_depth, self.max_depth + 1):
            c = self.channels(depth)
            with variable_scope(self._block_name("color", depth)):
                with variable_scope("conv"):
                    ops.get_weight([1, 1, 2, c], 2.0, True)
                    ops.get_bias([c])
            with variable_scope(self._block_name("conv", depth)):
                if depth == self.min_depth:
                    with variable_scope("conv"):
                        ops.get_weight([3, 3, c + 1, c], 2.0, True)
                        ops.get_bias([c])
                    with variable_scope("dense"):
                        ops.get_weight([c * int(self.resolution(depth).prod()), self.channels(depth - 1)], 2.0, True)
                        ops.get_bias([self.channels(depth - 1)])
                    with variable_scope("logits"):
                        ops.get_weight([self.channels(depth - 1), num_labels], 1.0, True)
                        ops.get_bias([num_labels])
                else:
                    with variable_scope("conv"):
                        ops.get_weight([3, 3, c, c], 2.0, True)
                        ops.get_bias([c])
                    with variable_scope("conv_downscale"):
                        ops.get_weight([3, 3, c, self.channels(depth - 1)], 2.0, True)
                        ops.get_bias([self.channels(depth - 1)])

    # The discriminator in two pieces (same variables, same launches as `discriminator`):
    #   trunk: colour block(s), the head block, the fade-in junction and the blocks down to `tail_top`
    #   tail:  the blocks from `tail_top` (at most 8x64) down to the 2x16 block with its statistic, dense and logits
    # Every launch of the tail is latency-bound at batch 8 (a few tens of blocks on 256 CUs), so the trainer runs the tails of the real and
    # of the fake pass of a discriminator run as ONE pass over the concatenated batch (models.GANSynth._d_losses_b): `sub_batches` keeps
    # the minibatch statistic per original batch (ops.py:336-348 on each half).
    TAIL_LEVELS = int(config.value("GS_D_TAIL_LEVELS", "3"))   # (measured: see DESIGN.md 6.3)

    def _tail_top(self, head):
        return max(self.min_depth, min(head - 1, self.min_depth + self.TAIL_LEVELS - 1))

    def discriminator_trunk(self, images, num_labels, name="discriminator", reuse=AUTO_REUSE):
        """-> (x, depth, fresh): the input of block `depth` (the first block of the tail)."""
        F.tap_begin("discriminator")
        with variable_scope(name, reuse=reuse):
            self._d_variables(num_labels)
            head, fade = self._head_depth(self.growing_depth)
            full = self.resolution(self.max_depth)

            def from_images(depth):
                return self._d_color_block(ops.downscale2d(images, full // self.resolution(depth)), depth)

            if head == self.min_depth:
                return from_images(head), head, False
            low = from_images(head - 1) if fade is not None else None   # lerp(low(), middle(), .): low first, like networks.py:271-275
            x = self._d_conv_block(from_images(head), head, num_labels, fresh_activation=True)
            fresh = fade is None
            if fade is not None:
                x = ops.lerp(low, x, fade if self.fade_weight is None else self.fade_weight)
            top = self._tail_top(head)
            for depth in range(head - 1, top, -1):
                x = self._d_conv_block(x, depth, num_labels, fresh_activation=fresh)
                fresh = True
            return x, top, fresh

    def discriminator_tail(self, x, depth, fresh, labels, sub_batches=1, name="discriminator", reuse=AUTO_REUSE):
        num_labels = labels.shape[1]
        with variable_scope(name, reuse=reuse):
            self._d_variables(num_labels)
            for d in range(depth, self.min_depth, -1):
                x = self._d_conv_block(x, d, num_labels, fresh_activation=fresh)
                fresh = True
            # (x also feeds batch_stddev in the last block: never fused)
            return self._d_conv_block(x, self.min_depth, num_labels, fresh_activation=False, sub_batches=sub_batches)

    def discriminator(self, images, labels, name="discriminator", reuse=AUTO_REUSE):
        x, depth, fresh = self.discriminator_trunk(images, labels.shape[1], name=name, reuse=reuse)
        return self.discriminator_tail(x, depth, fresh, labels, name=name, reuse=reuse)
