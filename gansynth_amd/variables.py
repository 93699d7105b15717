"""Name-scoped parameter store standing in for TF1's variable scopes.

The reference creates its parameters with tf.get_variable under nested tf.variable_scope
blocks (ops.py:156-180, networks.py:40-290) and shares them with reuse=tf.AUTO_REUSE.  The
store keeps the same names ("generator/conv_block_4x32/upscale_conv/weight", ...) and the same
layouts (conv HWIO, dense [in,out]), so a state dict is interchangeable with a TF checkpoint's
variable map.  All parameters are fp32 masters.
"""
import contextlib
import math
from collections import OrderedDict

import numpy as np
import torch

AUTO_REUSE = "auto_reuse"


class VariableStore(object):
    def __init__(self, device="cuda", seed=0):
        self.device = torch.device(device)
        self.variables = OrderedDict()
        self._scope = []
        self._gen = torch.Generator().manual_seed(seed)  # CPU generator: identical on every rank

    # -- scopes ---------------------------------------------------------------------------
    @contextlib.contextmanager
    def variable_scope(self, name, reuse=AUTO_REUSE):
        self._scope.append(name)
        try:
            yield
        finally:
            self._scope.pop()

    def full_name(self, name):
        return "/".join(self._scope + [name])

    # -- creation / lookup ----------------------------------------------------------------
    def get_variable(self, name, shape, initializer):
        full = self.full_name(name)
        var = self.variables.get(full)
        if var is None:
            value = initializer(tuple(int(s) for s in shape), self._gen)
            var = torch.nn.Parameter(value.to(device=self.device, dtype=torch.float32))
            self.variables[full] = var
        elif tuple(var.shape) != tuple(int(s) for s in shape):
            raise ValueError(f"variable {full} exists with shape {tuple(var.shape)}, requested {tuple(shape)}")
        return var

    def trainable_variables(self, scope):
        """tf.get_collection(TRAINABLE_VARIABLES, scope=...) (models.py:78-79)."""
        prefix = scope.rstrip("/") + "/"
        return OrderedDict((k, v) for k, v in self.variables.items() if k.startswith(prefix))

    # -- state ----------------------------------------------------------------------------
    def state_dict(self):
        return OrderedDict((k, v.detach().cpu().clone()) for k, v in self.variables.items())

    def load_state_dict(self, state, strict=True):
        with torch.no_grad():
            for k, t in state.items():
                if k in self.variables:
                    self.variables[k].copy_(torch.as_tensor(t).to(self.variables[k]))
                elif strict:
                    raise KeyError(f"unknown variable {k}")
        from . import kernels
        if kernels._K is not None and hasattr(kernels._K, "invalidate_weights"):
            kernels._K.invalidate_weights()
        missing = [k for k in self.variables if k not in state]
        if strict and missing:
            raise KeyError(f"state is missing {missing[:4]}...")


def truncated_normal(mean, stddev):
    """tf.initializers.truncated_normal: values beyond 2 sigma are re-drawn."""
    def init(shape, gen):
        t = torch.empty(shape, dtype=torch.float64)
        torch.nn.init.trunc_normal_(t, mean, stddev, mean - 2.0 * stddev, mean + 2.0 * stddev, generator=gen)
        return t.float()
    return init


def zeros():
    def init(shape, gen):
        return torch.zeros(shape, dtype=torch.float32)
    return init


_default = None


def default_store():
    global _default
    if _default is None:
        _default = VariableStore()
    return _default


def set_default_store(store):
    global _default
    _default = store
    return store


def variable_scope(name, reuse=AUTO_REUSE):
    return default_store().variable_scope(name, reuse)
