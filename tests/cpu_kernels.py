"""torch-CPU emulation of gansynth_amd.kernels.HipKernels -- TEST INFRASTRUCTURE ONLY.

It lets the CPU test-suite exercise the autograd algebra (functional.py), the op surface, the
network wiring and the trainer against the oracle without a GPU.  Each primitive is computed
independently of the HIP formulas (library convs / torch.autograd on the defining expression), so
agreement with the oracle checks the hand-derived backward / double-backward formulas too.
The product never imports this module.
"""
import torch
import torch.nn.functional as F

from oracle import torch_ref as R


def _lin_grad(fn, shape, like, gy):
    """Gradient of the linear map fn at any point, contracted with gy."""
    with torch.enable_grad():
        z = torch.zeros(shape, dtype=like.dtype, requires_grad=True)
        (g,) = torch.autograd.grad(fn(z), z, gy)
    return g


class CpuEmuKernels(object):
    lib = None
    _pending = None   # deferred in-place conv weight gradients (mirrors kernels.HipKernels.defer / flush_wgrad_reductions)

    def defer_wgrad_reductions(self):
        if self._pending is None:
            self._pending = []

    def flush_wgrad_reductions(self, group_of=None, on_group_done=None):
        pending, self._pending = self._pending, None
        if not pending:
            return 0
        if group_of is None:
            for _, _, fn in pending:
                fn()
            return len(pending)
        from gansynth_amd.kernels import completion_group
        tagged = {}
        for out, bias_out, fn in pending:
            tagged.setdefault(completion_group(group_of, out, bias_out), []).append(fn)
        for g in sorted(k for k in tagged if k is not None) + ([None] if None in tagged else []):
            for fn in tagged[g]:
                fn()
            if g is not None and on_group_done is not None:
                on_group_done(g)
        return len(pending)

    def invalidate_weights(self, flat=None):
        pass

    def refresh_weights(self, flat=None):
        return 0

    # conv2d (alpha folded, no bias)
    def conv2d_fwd(self, x, w, ksize, stride, alpha):
        w = w.detach().to(x.dtype)
        zero = torch.zeros(w.shape[3], dtype=x.dtype)
        scale = alpha / R.weight_scale(w.shape, 1.0)
        return (R.conv2d(x.detach(), w, zero, (stride, stride), 1.0) * scale).detach()

    def conv2d_fwd_bias_act(self, x, w, bias, ksize, stride, alpha, act):
        return self.bias_act_fwd(self.conv2d_fwd(x, w, ksize, stride, alpha), bias, act)

    def conv2d_transpose_fwd_bias_act(self, x, w, bias, alpha, act):
        return self.bias_act_fwd(self.conv2d_transpose_fwd(x, w, alpha), bias, act)

    def act_bwd_bias(self, g, y, act, out=None):
        gx = self.act_bwd(g, y, act)
        return gx, self.channel_sum(gx, out=out)

    def conv2d_bwd_data(self, gy, w, x_shape, ksize, stride, alpha, mask=None, mask_act=0):
        gx = _lin_grad(lambda z: self_conv(z, w, stride, alpha), tuple(x_shape), gy, gy.detach()).detach()
        return gx if mask is None else self.act_bwd(gx, mask, mask_act)

    @staticmethod
    def _out(val, out):
        if out is None:
            return val
        with torch.no_grad():
            out.add_(val.to(out.dtype))
        return out

    def conv2d_bwd_weight(self, x, gy, ksize, stride, alpha, out=None, bias_out=None):
        if out is not None and self._pending is not None:
            x, gy = x.detach(), gy.detach()
            pending, self._pending = self._pending, None
            pending.append((out, bias_out, lambda: self.conv2d_bwd_weight(x, gy, ksize, stride, alpha, out=out, bias_out=bias_out)))
            self._pending = pending
            return out
        ci, co = x.shape[1], gy.shape[1]
        shape = (ksize, ksize, ci, co)
        g = _lin_grad(lambda z: self_conv(x.detach(), z, stride, alpha), shape, gy, gy.detach())
        if bias_out is not None:
            self.channel_sum(gy, out=bias_out)
        return self._out(g.float().detach(), out)

    def conv2d_transpose_fwd(self, x, w, alpha):
        return self_convT(x.detach(), w.detach().to(x.dtype), alpha).detach()

    def conv2d_transpose_bwd_data(self, gy, w, alpha):
        n, co, h2, w2 = gy.shape
        ci = w.shape[2]
        return _lin_grad(lambda z: self_convT(z, w.detach().to(gy.dtype), alpha), (n, ci, h2 // 2, w2 // 2), gy, gy.detach()).detach()

    def conv2d_transpose_bwd_weight(self, x, gy, alpha, out=None):
        if out is not None and self._pending is not None:
            x, gy = x.detach(), gy.detach()
            pending, self._pending = self._pending, None
            pending.append((out, None, lambda: self.conv2d_transpose_bwd_weight(x, gy, alpha, out=out)))
            self._pending = pending
            return out
        shape = (3, 3, x.shape[1], gy.shape[1])
        return self._out(_lin_grad(lambda z: self_convT(x.detach(), z, alpha), shape, gy, gy.detach()).float().detach(), out)

    def dense_fwd(self, x, w, alpha):
        return (x.detach() @ w.detach().to(x.dtype)) * alpha

    def dense_bwd_data(self, gy, w, alpha):
        return (gy.detach() @ w.detach().to(gy.dtype).t()) * alpha

    def dense_bwd_weight(self, x, gy, alpha, out=None):
        return self._out(((x.detach().t() @ gy.detach()) * alpha).float(), out)

    def embedding_fwd(self, idx, w, alpha, dtype):
        return (w.detach()[idx] * alpha).to(dtype)

    def embedding_bwd(self, idx, gy, rows, alpha):
        gw = torch.zeros(rows, gy.shape[1], dtype=torch.float32)
        gw.index_add_(0, idx, gy.detach().float())
        return gw * alpha

    def bias_act_fwd(self, x, bias, act):
        y = x.detach()
        if bias is not None:
            b = bias.detach().to(x.dtype)
            y = y + (b.view(1, -1, 1, 1) if x.dim() == 4 else b.view(1, -1))
        if act == 1:
            y = F.leaky_relu(y, 0.2)
        elif act == 2:
            y = torch.tanh(y)
        return y

    def act_bwd(self, g, y, act):
        g, y = g.detach(), y.detach()
        if act == 1:
            return torch.where(y > 0, g, 0.2 * g)
        return g * (1 - y * y)

    def tanh_bwd_bwd(self, gg, g, y):
        return -2.0 * y.detach() * g.detach() * gg.detach()

    def channel_sum(self, g, out=None):
        g = g.detach().float()
        return self._out(g.sum(dim=(0, 2, 3)) if g.dim() == 4 else g.sum(dim=0), out)

    def pixel_norm_fwd(self, x, eps):
        return R.pixel_normalization(x.detach(), eps)

    def pixel_norm_bwd(self, g, x, eps, act=0, pre_act=0, addend=None):
        g = g.detach().expand_as(x)
        if pre_act:
            g = self.act_bwd(g, x, pre_act)
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(R.pixel_normalization(xx, eps), xx, g)
        if addend is not None:
            gx = gx + addend.detach()
        return gx if act == 0 else self.act_bwd(gx, x, act)

    def pixel_norm_bwd_bwd(self, gg, g, x, eps, pre_act=0, with_g=False):
        if with_g:
            return self.pixel_norm_bwd_bwd(gg, g, x, eps, pre_act=pre_act), self.pixel_norm_bwd(gg, x, eps, pre_act=pre_act)
        if pre_act:
            gg = self.act_bwd(gg.detach().expand_as(x), x, pre_act)
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(R.pixel_normalization(xx, eps), xx, g.detach().expand_as(xx), create_graph=True)
            (out,) = torch.autograd.grad(gx, xx, gg.detach())
        return out

    def upscale2d(self, x, fy, fx, scale):
        return R.upscale2d(x.detach(), (fy, fx)) * scale

    def blocksum2d(self, x, fy, fx, scale):
        return R.downscale2d(x.detach(), (fy, fx)) * (fy * fx * scale)

    def batch_stddev_fwd(self, x, eps):
        return R.batch_stddev(x.detach(), 4, eps)

    def batch_stddev_bwd(self, gy, x, eps, addend=None):
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(R.batch_stddev(xx, 4, eps), xx, gy.detach())
        return gx if addend is None else gx + addend.detach()

    def batch_stddev_bwd_bwd(self, ggx, gy, x, eps):
        with torch.enable_grad():
            xx = x.detach().clone().requires_grad_(True)
            gyy = gy.detach().clone().requires_grad_(True)
            (gx,) = torch.autograd.grad(R.batch_stddev(xx, 4, eps), xx, gyy, create_graph=True)
            ggy, gx2 = torch.autograd.grad(gx, [gyy, xx], ggx.detach())
        return ggy, gx2

    def axpby(self, a, b, ca, cb):
        return ca * a.detach() + cb * b.detach().expand_as(a)

    def sumsq_rows(self, x):
        return x.detach().float().pow(2).reshape(x.shape[0], -1).sum(dim=1)

    def row_scale(self, x, s, alpha=1.0):
        return x.detach() * (alpha * s.detach()).to(x.dtype).view(-1, *([1] * (x.dim() - 1)))

    def adam_tf_step(self, p, g, m, v, lr_t, beta1, beta2, eps, grad_scale=1.0, refresh=True, zero_grad=False):
        with torch.no_grad():
            gr = g * grad_scale
            m.mul_(beta1).add_(gr, alpha=1 - beta1)
            v.mul_(beta2).addcmul_(gr, gr, value=1 - beta2)
            p.sub_(lr_t * m / (v.sqrt() + eps))
            if zero_grad:
                g.zero_()


def self_conv(x, w, stride, alpha):
    w = w.to(x.dtype)
    zero = torch.zeros(w.shape[3], dtype=x.dtype)
    return R.conv2d(x, w, zero, (stride, stride), 1.0) * (alpha / R.weight_scale(w.shape, 1.0))


def self_convT(x, w, alpha):
    w = w.to(x.dtype)
    zero = torch.zeros(w.shape[3], dtype=x.dtype)
    return R.conv2d_transpose(x, w, zero, (2, 2), 1.0) * (alpha / R.weight_scale(w.shape, 1.0))
