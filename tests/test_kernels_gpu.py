"""GPU: every HIP entry point (called through the C ABI via gansynth_amd.kernels) against an
independent torch-CPU computation of the same primitive (tests/cpu_kernels.py, which is built on the
oracle's restatements).  Tolerance: 1e-3 relative (BASELINE.json north_star) on the tensor scale;
index/copy ops are bit-exact."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def K():
    from gansynth_amd import kernels
    return kernels.HipKernels()


@pytest.fixture(scope="module")
def E():
    from tests.cpu_kernels import CpuEmuKernels
    return CpuEmuKernels()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def dev(t, dtype=torch.float32):
    t = t.to("cuda", dtype)
    return t.contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t


def close(got, ref, rel=1e-3, name=""):
    got = got.detach().float().cpu()
    ref = ref.detach().float()
    assert got.shape == ref.shape, f"{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}"
    scale = float(ref.abs().max()) + 1e-30
    err = float((got - ref).abs().max()) / scale
    assert err <= rel, f"{name}: max err / max|ref| = {err:.3e} > {rel:.1e}"


# (n, ci, co, h, w, ksize, stride) -- covers every MFMA tile configuration + the direct kernels
CONV_CASES = [
    (2, 32, 32, 8, 128, 3, 1),    # S1 A1 B2 TW64 TG9 (top-level shape class)
    (2, 32, 32, 4, 32, 3, 1),     # S1 A1 B1 TW32
    (1, 64, 64, 8, 64, 3, 1),     # S1 A2 B2 TW64
    (2, 64, 64, 4, 32, 3, 1),     # S1 A2 B1 TW32
    (4, 64, 64, 2, 16, 3, 1),     # S1 A2 B1 TW16 (tile taller than the image)
    (1, 128, 128, 8, 32, 3, 1),   # S1 A4 B1 TW32
    (4, 256, 256, 2, 16, 3, 1),   # S1 A4 B1 TW16, two oc tiles (the 2x16 stage)
    (2, 64, 32, 6, 40, 3, 1),     # ragged spatial tile edges, ci != co
    (2, 32, 64, 8, 128, 3, 2),    # S2 A2 B1 TW32  (D downscale 32->64)
    (2, 64, 128, 8, 64, 3, 2),    # S2 A4 B1 TW32
    (4, 256, 256, 4, 32, 3, 2),   # S2 A4 TW16 (4x32 -> 2x16)
    (2, 64, 32, 8, 64, 3, 2),     # S2 A1 (oc 32); its bwd-data is T2 with 64 outputs
    (2, 32, 32, 12, 72, 3, 2),    # ragged stride-2
    (2, 32, 2, 8, 64, 1, 1),      # G colour block 32->2 (direct)
    (2, 2, 32, 8, 64, 1, 1),      # D colour block 2->32 (direct)
    (4, 1, 16, 2, 16, 3, 1),      # minibatch-stddev plane (direct, ci=1)
    (2, 16, 2, 4, 32, 1, 1),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_three_maps(K, E, case):
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1)
    wt = rnd(ks, ks, ci, co, seed=2)
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    y_ref = E.conv2d_fwd(x, wt, ks, st, alpha)
    close(K.conv2d_fwd(dev(x), dev(wt), ks, st, alpha), y_ref, name="fwd")
    gy = rnd(*y_ref.shape, seed=3)
    close(K.conv2d_bwd_data(dev(gy), dev(wt), x.shape, ks, st, alpha), E.conv2d_bwd_data(gy, wt, x.shape, ks, st, alpha), name="bwd_data")
    close(K.conv2d_bwd_weight(dev(x), dev(gy), ks, st, alpha), E.conv2d_bwd_weight(x, gy, ks, st, alpha), name="bwd_weight")


CONVT_CASES = [
    (2, 64, 32, 8, 64),    # T2 A1 B2 TW64 (top level 64->32)
    (2, 64, 32, 4, 32),    # T2 A1 B1 TW32
    (1, 128, 64, 8, 32),   # T2 A2 B1 TW32
    (4, 256, 256, 2, 16),  # T2 A2 B1 TW16, 4 oc tiles (2x16 -> 4x32)
    (2, 32, 64, 5, 24),    # ragged, ci < co
]


@pytest.mark.parametrize("case", CONVT_CASES)
def test_conv2d_transpose_three_maps(K, E, case):
    n, ci, co, h, w = case
    x = rnd(n, ci, h, w, seed=4)
    wt = rnd(3, 3, ci, co, seed=5)
    alpha = float(np.sqrt(2.0 / (9 * ci)))
    y_ref = E.conv2d_transpose_fwd(x, wt, alpha)
    assert y_ref.shape == (n, co, 2 * h, 2 * w)
    close(K.conv2d_transpose_fwd(dev(x), dev(wt), alpha), y_ref, name="fwd")
    gy = rnd(*y_ref.shape, seed=6)
    close(K.conv2d_transpose_bwd_data(dev(gy), dev(wt), alpha), E.conv2d_transpose_bwd_data(gy, wt, alpha), name="bwd_data")
    close(K.conv2d_transpose_bwd_weight(dev(x), dev(gy), alpha), E.conv2d_transpose_bwd_weight(x, gy, alpha), name="bwd_weight")


@pytest.mark.parametrize("b,i,o", [(8, 512, 8192), (8, 8192, 256), (8, 256, 61), (4, 32, 40), (20, 100, 70), (16, 8192, 256), (24, 8192, 256), (13, 512, 8192),
                                   (24, 512, 8192), (30, 8192, 256)])
def test_dense(K, E, b, i, o):
    x, w, gy = rnd(b, i, seed=1), rnd(i, o, seed=2), rnd(b, o, seed=3)
    alpha = float(np.sqrt(2.0 / i))
    close(K.dense_fwd(dev(x), dev(w), alpha), E.dense_fwd(x, w, alpha), name="fwd")
    close(K.dense_bwd_data(dev(gy), dev(w), alpha), E.dense_bwd_data(gy, w, alpha), name="bwd_data")
    close(K.dense_bwd_weight(dev(x), dev(gy), alpha), E.dense_bwd_weight(x, gy, alpha), name="bwd_weight")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b,i,o,act", [(8, 8192, 256, 1), (16, 8192, 256, 1), (16, 256, 61, 0), (8, 512, 8192, 1), (5, 100, 70, 2), (24, 8192, 256, 1), (8, 256, 61, 0)])
def test_dense_with_bias_and_activation_where_the_forward_writes(K, E, b, i, o, act, dtype):
    """gs_dense_fwd_bias_act (ops.py:183-201: matmul, bias_add, activation as the reference calls them -- the discriminator's dense + leaky_relu
    and its logits layer): against the oracle-side emulation, and in fp32 bit-identical to the two separate entry points (same operations,
    same order) on every kernel path (MFMA split-K + finalize, one-launch small layer, wide direct, generic)."""
    x, w, bias = rnd(b, i, seed=1).to(dtype).float(), rnd(i, o, seed=2), rnd(o, seed=3)
    alpha = float(np.sqrt(2.0 / i))
    want = E.bias_act_fwd(E.dense_fwd(x, w, alpha), bias, act)
    got = K.dense_fwd_bias_act(dev(x, dtype), dev(w), dev(bias), alpha, act)
    close(got, want, rel=2e-2 if dtype == torch.bfloat16 else 1e-3, name="fused")
    two = K.bias_act_fwd(K.dense_fwd(dev(x, dtype), dev(w), alpha), dev(bias), act)
    if dtype == torch.float32:
        assert torch.equal(got, two)
    else:   # one rounding instead of two: at least as close to the fp32 evaluation
        ref = K.bias_act_fwd(K.dense_fwd(dev(x), dev(w), alpha), dev(bias), act).double()
        assert float((got.double() - ref).pow(2).mean()) <= 1.05 * float((two.double() - ref).pow(2).mean()) + 1e-12
    assert torch.equal(K.dense_fwd_bias_act(dev(x, dtype), dev(w), None, alpha, 0), K.dense_fwd(dev(x, dtype), dev(w), alpha))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("b,c,h,w,o", [(8, 256, 2, 16, 256), (4, 64, 2, 16, 256), (16, 256, 2, 16, 512)])
def test_dense_of_a_flattened_channels_last_activation(K, E, b, c, h, w, o, dtype):
    """gs_dense_*_nhwc: dense(tf.layers.flatten(x)) with x left in channels-last memory (the flatten is a row map inside the kernels)
    -- all three maps against the plain entry points on the explicitly flattened (channel-major) copy."""
    x4 = rnd(b, c, h, w, seed=1).to(dtype).float()
    wt, gy = rnd(c * h * w, o, seed=2), rnd(b, o, seed=3).to(dtype).float()
    alpha = float(np.sqrt(2.0 / (c * h * w)))
    x_cl = dev(x4, dtype)                      # channels-last device tensor
    assert K.dense_nhwc_ok(x_cl, o)
    flat = x4.reshape(b, -1)                   # NCHW flatten: column c * hw + p
    close(K.dense_fwd_nhwc(x_cl, dev(wt), alpha), E.dense_fwd(flat, wt, alpha), rel=2e-2 if dtype == torch.bfloat16 else 1e-3, name="fwd")
    bias = rnd(o, seed=4)
    close(K.dense_fwd_bias_act(x_cl, dev(wt), dev(bias), alpha, 1), E.bias_act_fwd(E.dense_fwd(flat, wt, alpha), bias, 1),
          rel=2e-2 if dtype == torch.bfloat16 else 1e-3, name="fwd + bias + leaky_relu")
    gx = K.dense_bwd_data_nhwc(dev(gy, dtype), dev(wt), (b, c, h, w), alpha)
    assert gx.shape == (b, c, h, w) and gx.is_contiguous(memory_format=torch.channels_last)
    close(gx.float().cpu().reshape(b, -1), E.dense_bwd_data(gy, wt, alpha), rel=2e-2 if dtype == torch.bfloat16 else 1e-3, name="bwd_data")
    close(K.dense_bwd_weight_nhwc(x_cl, dev(gy, dtype), alpha), E.dense_bwd_weight(flat, gy, alpha), name="bwd_weight")
    acc = torch.full((c * h * w, o), 0.5, device="cuda")
    K.dense_bwd_weight_nhwc(x_cl, dev(gy, dtype), alpha, out=acc)
    close(acc, E.dense_bwd_weight(flat, gy, alpha) + 0.5, name="bwd_weight +=")


def test_embedding_exact(K, E):
    w = rnd(61, 256, seed=1)
    idx = torch.tensor([3, 60, 0, 3, 17, 3, 59, 1])
    got = K.embedding_fwd(idx.cuda(), dev(w), 0.5, torch.float32).cpu()
    assert torch.equal(got, E.embedding_fwd(idx, w, 0.5, torch.float32))  # one multiply per element: bit-exact
    gy = rnd(8, 256, seed=2)
    close(K.embedding_bwd(idx.cuda(), dev(gy), 61, 0.5), E.embedding_bwd(idx, gy, 61, 0.5), rel=1e-6, name="bwd")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_embedding_of_one_hot_rows_in_one_launch(K, E, dtype):
    """gs_embedding_onehot_fwd: the row index comes from the one-hot input itself (ops.py:207, tf.argmax: the first maximum) --
    same rows, same bits as argmax + gs_embedding_fwd, and the index it hands to the backward."""
    w = rnd(61, 256, seed=1)
    idx = torch.tensor([3, 60, 0, 3, 17, 3, 59, 1])
    labels = torch.nn.functional.one_hot(idx, 61).to(dtype)
    y, got_idx = K.embedding_onehot_fwd(labels.cuda(), dev(w), 0.5)
    assert torch.equal(got_idx.cpu(), idx)
    assert torch.equal(y.cpu(), K.embedding_fwd(idx.cuda(), dev(w), 0.5, dtype).cpu())
    soft = rnd(8, 61, seed=2)
    soft[2, 7] = soft[2, 40] = soft[2].max() + 1.0   # a tie: the first maximum wins
    _, i2 = K.embedding_onehot_fwd(soft.to(dtype).cuda(), dev(w), 0.5)
    assert int(i2[2]) == 7 and torch.equal(i2.cpu(), torch.argmax(soft.to(dtype).float(), dim=1)) or int(i2[2]) == 7


@pytest.mark.parametrize("shape", [(2, 32, 8, 64), (4, 256, 2, 16), (2, 2, 16, 128), (8, 8192), (8, 61), (3, 64, 5, 7)])
@pytest.mark.parametrize("act", [0, 1, 2])
def test_bias_act_and_grads(K, E, shape, act):
    x, g, gg = rnd(*shape, seed=1), rnd(*shape, seed=2), rnd(*shape, seed=3)
    bias = rnd(shape[1], seed=4)
    y_ref = E.bias_act_fwd(x, bias, act)
    y = K.bias_act_fwd(dev(x), dev(bias), act)
    close(y, y_ref, rel=1e-5, name="fwd")
    close(K.bias_act_fwd(dev(x), None, act), E.bias_act_fwd(x, None, act), rel=1e-5, name="fwd_nobias")
    if act:
        close(K.act_bwd(dev(g), dev(y_ref), act), E.act_bwd(g, y_ref, act), rel=1e-5, name="act_bwd")
    if act == 2:
        close(K.tanh_bwd_bwd(dev(gg), dev(g), dev(y_ref)), E.tanh_bwd_bwd(gg, g, y_ref), rel=1e-5, name="tanh_bwd_bwd")
    close(K.channel_sum(dev(g)), E.channel_sum(g), rel=1e-5, name="channel_sum")


def test_channel_sum_large(K, E):
    g = rnd(2, 32, 128, 256, seed=5)
    close(K.channel_sum(dev(g)), E.channel_sum(g), rel=1e-5)


@pytest.mark.parametrize("shape", [(2, 32, 8, 64), (4, 256, 2, 16), (3, 64, 5, 9), (2, 128, 4, 16), (8, 512), (4, 8, 3, 5)])
def test_pixel_norm_all_orders(K, E, shape):
    x, g, gg = rnd(*shape, seed=1), rnd(*shape, seed=2), rnd(*shape, seed=3)
    eps = 1e-12
    close(K.pixel_norm_fwd(dev(x), eps), E.pixel_norm_fwd(x, eps), rel=1e-5, name="fwd")
    close(K.pixel_norm_bwd(dev(g), dev(x), eps), E.pixel_norm_bwd(g, x, eps), rel=1e-4, name="bwd")
    close(K.pixel_norm_bwd_bwd(dev(gg), dev(g), dev(x), eps), E.pixel_norm_bwd_bwd(gg, g, x, eps), rel=1e-4, name="bwd_bwd")


@pytest.mark.parametrize("n,c,h,w,fy,fx", [(2, 2, 2, 16, 64, 64), (2, 2, 8, 64, 16, 16), (2, 2, 64, 512, 2, 2), (1, 4, 4, 4, 3, 2), (2, 32, 4, 8, 2, 2)])
def test_upscale_exact_and_blocksum(K, E, n, c, h, w, fy, fx):
    x = rnd(n, c, h, w, seed=1)
    up = K.upscale2d(dev(x), fy, fx, 1.0).cpu()
    assert torch.equal(up, E.upscale2d(x, fy, fx, 1.0))  # index op: bit-exact
    big = rnd(n, c, h * fy, w * fx, seed=2)
    close(K.blocksum2d(dev(big), fy, fx, 1.0 / (fy * fx)), E.blocksum2d(big, fy, fx, 1.0 / (fy * fx)), rel=1e-5, name="avgpool")
    close(K.blocksum2d(dev(big), fy, fx, 1.0), E.blocksum2d(big, fy, fx, 1.0), rel=1e-5, name="blocksum")


@pytest.mark.parametrize("b,c,h,w", [(4, 256, 2, 16), (8, 256, 2, 16), (8, 16, 3, 5)])
def test_batch_stddev_all_orders(K, E, b, c, h, w):
    x = rnd(b, c, h, w, seed=1)
    gy = rnd(b, 1, h, w, seed=2)
    ggx = rnd(b, c, h, w, seed=3)
    eps = 1e-12
    close(K.batch_stddev_fwd(dev(x), eps), E.batch_stddev_fwd(x, eps), rel=1e-5, name="fwd")
    close(K.batch_stddev_bwd(dev(gy), dev(x), eps), E.batch_stddev_bwd(gy, x, eps), rel=1e-4, name="bwd")
    close(K.batch_stddev_bwd(dev(gy), dev(x), eps, addend=dev(0.5 * x)), E.batch_stddev_bwd(gy, x, eps, addend=0.5 * x), rel=1e-4, name="bwd + addend")
    ggy, gx2 = K.batch_stddev_bwd_bwd(dev(ggx), dev(gy), dev(x), eps)
    rggy, rgx2 = E.batch_stddev_bwd_bwd(ggx, gy, x, eps)
    close(ggy, rggy, rel=1e-4, name="bwd_bwd.ggy")
    close(gx2, rgx2, rel=1e-4, name="bwd_bwd.gx")


def test_axpby_sumsq_rowscale(K, E):
    a, b = rnd(2, 2, 16, 128, seed=1), rnd(2, 2, 16, 128, seed=2)
    close(K.axpby(dev(a), dev(b), 0.3, 0.7), E.axpby(a, b, 0.3, 0.7), rel=1e-6)
    close(K.axpby(dev(a[:, :, :3, :5]), dev(b[:, :, :3, :5]), -1.0, 2.0), E.axpby(a[:, :, :3, :5], b[:, :, :3, :5], -1.0, 2.0), rel=1e-6)
    x = rnd(8, 2, 32, 64, seed=3)
    close(K.sumsq_rows(dev(x)), E.sumsq_rows(x), rel=1e-5)
    s = rnd(8, seed=4)
    close(K.row_scale(dev(x), dev(s)), E.row_scale(x, s), rel=1e-6)
    close(K.row_scale(dev(x), dev(s), 2.0), 2.0 * E.row_scale(x, s), rel=1e-6)


def test_adam_tf_step(K, E):
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    m, v = rnd(n, seed=3).abs() * 0.1, rnd(n, seed=4).abs() * 0.1
    rp, rm, rv = p.clone(), m.clone(), v.clone()
    E.adam_tf_step(rp, g, rm, rv, 1e-3, 0.0, 0.99, 1e-8, 0.5)
    dp, dg, dm, dv = p.cuda(), g.cuda(), m.cuda(), v.cuda()
    K.adam_tf_step(dp, dg, dm, dv, 1e-3, 0.0, 0.99, 1e-8, 0.5)
    close(dp, rp, rel=1e-6)
    close(dm, rm, rel=1e-6)
    close(dv, rv, rel=1e-6)


@pytest.mark.parametrize("zero_grad", [False, True])
def test_adam_tf_step_with_lr_from_device_memory(K, E, zero_grad):
    """gs_adam_tf_step_dev (the optimizer step as a node of the iteration's hipGraph): lr_t is read from device memory WHEN THE LAUNCH RUNS --
    bit-identical to the by-value entry points for the same fp32 lr_t, against the oracle-side TF-Adam within fp32 round-off, a negative
    value leaves every buffer untouched, and a value written after capture is the one a replay uses."""
    from gansynth_amd import functional as F
    n = 100003
    p, g = rnd(n, seed=1), rnd(n, seed=2)
    m, v = rnd(n, seed=3).abs() * 0.1, rnd(n, seed=4).abs() * 0.1
    lr_t = 8e-4 * np.sqrt(1.0 - 0.99 ** 3) / (1.0 - 0.0 ** 3)
    rp, rm, rv = p.clone(), m.clone(), v.clone()
    E.adam_tf_step(rp, g, rm, rv, lr_t, 0.0, 0.99, 1e-8, 0.5)
    by_value = [t.cuda() for t in (p, g, m, v)]
    K.adam_tf_step(*by_value, lr_t, 0.0, 0.99, 1e-8, 0.5, refresh=False, zero_grad=zero_grad)
    table = F.DeviceScalars("cuda", 2)
    table.set([lr_t, -1.0])
    dev_ = [t.cuda() for t in (p, g, m, v)]
    K.adam_tf_step_dev(*dev_, table.ptr(0), 0.0, 0.99, 1e-8, 0.5, refresh=False, zero_grad=zero_grad)
    for a, b, name in zip(dev_, by_value, "pgmv"):
        assert torch.equal(a, b), name                      # same arithmetic on the same fp32 scalar
    close(dev_[0], rp, rel=1e-6)
    close(dev_[2], rm, rel=1e-6)
    close(dev_[3], rv, rel=1e-6)
    assert bool((dev_[1] == 0).all()) == zero_grad
    skipped = [t.cuda() for t in (p, g, m, v)]
    K.adam_tf_step_dev(*skipped, table.ptr(1), 0.0, 0.99, 1e-8, 0.5, refresh=False, zero_grad=True)   # negative lr_t: no step pending
    for a, b in zip(skipped, (p, g, m, v)):
        assert torch.equal(a.cpu(), b)
    # inside a captured graph the scalar is read at replay time
    cap = [t.cuda() for t in (p, g, m, v)]
    table.set([-1.0, -1.0])
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        K.adam_tf_step_dev(*cap, table.ptr(0), 0.0, 0.99, 1e-8, 0.5, refresh=False, zero_grad=zero_grad)
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(cap[0].cpu(), p)                     # captured with "no step": nothing moved
    table.set([lr_t, -1.0])
    graph.replay()
    torch.cuda.synchronize()
    for a, b, name in zip(cap, by_value, "pgmv"):
        assert torch.equal(a, b), name


@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 4, 32, 3, 1), (2, 32, 64, 8, 128, 3, 2), (4, 256, 256, 2, 16, 3, 1)])
def test_conv2d_bf16_forward_and_data_grad(K, E, case):
    """bf16 storage / fp32 accumulate path of the MFMA kernels against the fp32 reference on bf16-rounded inputs."""
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1).bfloat16().float()
    wt = rnd(ks, ks, ci, co, seed=2)
    wt_r = wt.bfloat16().float()
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    y_ref = E.conv2d_fwd(x, wt_r, ks, st, alpha)
    close(K.conv2d_fwd(dev(x, torch.bfloat16), dev(wt), ks, st, alpha), y_ref, rel=1e-2, name="fwd")
    gy = rnd(*y_ref.shape, seed=3).bfloat16().float()
    close(K.conv2d_bwd_data(dev(gy, torch.bfloat16), dev(wt), x.shape, ks, st, alpha), E.conv2d_bwd_data(gy, wt_r, x.shape, ks, st, alpha), rel=1e-2, name="bwd_data")


@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 32, 64, 8, 128, 3, 2), (4, 256, 256, 2, 16, 3, 1)])
def test_conv2d_bf16_weight_grad(K, E, case):
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1).bfloat16().float()
    gy = rnd(n, co, h // st, w // st, seed=3).bfloat16().float()
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    close(K.conv2d_bwd_weight(dev(x, torch.bfloat16), dev(gy, torch.bfloat16), ks, st, alpha), E.conv2d_bwd_weight(x, gy, ks, st, alpha), rel=1e-4, name="bwd_weight")


def test_conv2d_transpose_bf16(K, E):
    n, ci, co, h, w = 2, 64, 32, 8, 64
    x = rnd(n, ci, h, w, seed=4).bfloat16().float()
    wt = rnd(3, 3, ci, co, seed=5)
    alpha = float(np.sqrt(2.0 / (9 * ci)))
    close(K.conv2d_transpose_fwd(dev(x, torch.bfloat16), dev(wt), alpha), E.conv2d_transpose_fwd(x, wt.bfloat16().float(), alpha), rel=1e-2, name="fwd")
    gy = rnd(n, co, 2 * h, 2 * w, seed=6).bfloat16().float()
    close(K.conv2d_transpose_bwd_weight(dev(x, torch.bfloat16), dev(gy, torch.bfloat16), alpha), E.conv2d_transpose_bwd_weight(x, gy, alpha), rel=1e-4, name="bwd_weight")


@pytest.mark.parametrize("n,ci,co,h,w", [(2, 32, 2, 64, 256), (2, 2, 32, 64, 256), (4, 256, 2, 2, 16), (4, 2, 256, 2, 16), (3, 64, 2, 7, 9)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_thin_colour_weight_grad(K, E, n, ci, co, h, w, dtype):
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    gy = rnd(n, co, h, w, seed=3).to(dtype).float()
    close(K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), 1, 1, 0.25), E.conv2d_bwd_weight(x, gy, 1, 1, 0.25), rel=1e-4, name="thin wgrad")


@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 4, 32, 3, 1), (2, 32, 64, 8, 128, 3, 2), (2, 32, 2, 8, 64, 1, 1), (4, 256, 256, 2, 16, 3, 1)])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv2d_fused_bias_act_epilogue(K, E, case, act, dtype):
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    wt = rnd(ks, ks, ci, co, seed=2)
    wr = wt.to(dtype).float() if (dtype == torch.bfloat16 and ks == 3) else wt
    bias = rnd(co, seed=7)
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    ref = E.conv2d_fwd_bias_act(x, wr, bias, ks, st, alpha, act)
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    close(K.conv2d_fwd_bias_act(dev(x, dtype), dev(wt), dev(bias), ks, st, alpha, act), ref, rel=tol, name="fused fwd")
    close(K.conv2d_fwd_bias_act(dev(x, dtype), dev(wt), None, ks, st, alpha, act), E.conv2d_fwd_bias_act(x, wr, None, ks, st, alpha, act), rel=tol, name="fused fwd nobias")


@pytest.mark.parametrize("act", [0, 1])
def test_conv2d_transpose_fused_bias_act(K, E, act):
    n, ci, co, h, w = 2, 64, 32, 8, 64
    x, wt, bias = rnd(n, ci, h, w, seed=4), rnd(3, 3, ci, co, seed=5), rnd(co, seed=6)
    close(K.conv2d_transpose_fwd_bias_act(dev(x), dev(wt), dev(bias), 0.1, act), E.conv2d_transpose_fwd_bias_act(x, wt, bias, 0.1, act), name="convT fused")


@pytest.mark.parametrize("shape", [(2, 32, 64, 256), (4, 256, 2, 16), (2, 2, 16, 128), (8, 8192), (3, 64, 5, 7), (8, 61)])
@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_act_bwd_bias_fused(K, E, shape, act, dtype):
    g = rnd(*shape, seed=2).to(dtype).float()
    y = E.bias_act_fwd(rnd(*shape, seed=1), None, act).to(dtype).float()
    gx_ref, gb_ref = E.act_bwd_bias(g, y, act)
    gx, gb = K.act_bwd_bias(dev(g, dtype), dev(y, dtype), act)
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    close(gx, gx_ref, rel=tol, name="gx")
    # the sum is taken over the unrounded fp32 values in registers: compare against the exact sum
    close(gb, gb_ref, rel=1e-4 if dtype == torch.float32 else 2e-2, name="gb")


def test_prepared_weight_operand_cache(K, E):
    """A registered parameter keeps its re-laid operand between calls until its values change."""
    flat = torch.randn(9 * 32 * 32 + 64, device="cuda")
    K.register_param_buffer(flat)
    w = flat[:9 * 32 * 32].view(3, 3, 32, 32)
    x = rnd(2, 32, 8, 64, seed=1)
    ref1 = E.conv2d_fwd(x, w.cpu(), 3, 1, 0.1)
    close(K.conv2d_fwd(dev(x), w, 3, 1, 0.1), ref1, name="first call (prepares)")
    close(K.conv2d_fwd(dev(x), w, 3, 1, 0.1), ref1, name="second call (reuses)")
    close(K.conv2d_bwd_data(dev(ref1), w, x.shape, 3, 1, 0.1), E.conv2d_bwd_data(ref1, w.cpu(), x.shape, 3, 1, 0.1), name="bwd_data")
    assert any(k[0] == w.data_ptr() for k in K._wcache)
    # raw-pointer update (what the Adam kernel does) + explicit invalidation
    K.adam_tf_step(flat, torch.ones_like(flat), torch.zeros_like(flat), torch.zeros_like(flat), 0.5, 0.0, 0.99, 1e-8, 1.0)
    ref2 = E.conv2d_fwd(x, w.cpu(), 3, 1, 0.1)
    assert float((ref2 - ref1).abs().max()) > 1e-3
    close(K.conv2d_fwd(dev(x), w, 3, 1, 0.1), ref2, name="after the parameter changed")
    # in-place torch update bumps the tensor version
    with torch.no_grad():
        w.mul_(0.5)
    close(K.conv2d_fwd(dev(x), w, 3, 1, 0.1), E.conv2d_fwd(x, w.cpu(), 3, 1, 0.1), name="after an in-place torch update")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_parameter_gradients_accumulate_in_kernel(K, E, dtype):
    """`out=`: the gradient is added into an existing fp32 buffer by the reducing kernel itself (accumulate=1 in the
    C ABI) -- what functional.py uses to write straight into the flat gradient buffer."""
    tol = 1e-4
    for (n, ci, co, h, w, ks, st) in [(2, 32, 32, 8, 128, 3, 1), (2, 32, 64, 8, 128, 3, 2), (2, 32, 2, 8, 64, 1, 1), (4, 1, 16, 2, 16, 3, 1)]:
        x = rnd(n, ci, h, w, seed=1).to(dtype).float()
        gy = rnd(n, co, h // st, w // st, seed=2).to(dtype).float()
        base = rnd(ks, ks, ci, co, seed=3)
        acc = dev(base).contiguous()
        got = K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), ks, st, 0.2, out=acc)
        assert got.data_ptr() == acc.data_ptr()
        close(acc, base + E.conv2d_bwd_weight(x, gy, ks, st, 0.2), rel=tol, name=f"conv wgrad += {ci}->{co} k{ks} s{st}")
    x, gy, base = rnd(2, 64, 8, 64, seed=4).to(dtype).float(), rnd(2, 32, 16, 128, seed=5).to(dtype).float(), rnd(3, 3, 64, 32, seed=6)
    acc = dev(base).contiguous()
    K.conv2d_transpose_bwd_weight(dev(x, dtype), dev(gy, dtype), 0.1, out=acc)
    close(acc, base + E.conv2d_transpose_bwd_weight(x, gy, 0.1), rel=tol, name="convT wgrad +=")
    x, gy, base = rnd(8, 512, seed=7).to(dtype).float(), rnd(8, 256, seed=8).to(dtype).float(), rnd(512, 256, seed=9)
    acc = dev(base)
    K.dense_bwd_weight(dev(x, dtype), dev(gy, dtype), 0.3, out=acc)
    close(acc, base + E.dense_bwd_weight(x, gy, 0.3), rel=tol, name="dense wgrad +=")
    for shape in [(2, 32, 64, 256), (8, 8192), (3, 64, 5, 7)]:
        g = rnd(*shape, seed=2).to(dtype).float()
        y = E.bias_act_fwd(rnd(*shape, seed=1), None, 1).to(dtype).float()
        base = rnd(shape[1], seed=3)
        acc = dev(base)
        K.channel_sum(dev(g, dtype), out=acc)
        close(acc, base + E.channel_sum(g), rel=tol, name="channel_sum +=")
        acc = dev(base)
        K.act_bwd_bias(dev(g, dtype), dev(y, dtype), 1, out=acc)
        close(acc, base + E.act_bwd_bias(g, y, 1)[1], rel=1e-4 if dtype == torch.float32 else 2e-2, name="act_bwd_bias +=")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 8, 64, 3, 1), (2, 64, 128, 8, 64, 3, 2), (2, 32, 64, 8, 128, 3, 2),
                                  (4, 256, 256, 2, 16, 3, 1), (2, 2, 32, 8, 64, 1, 1)])
def test_bias_gradient_rides_with_weight_gradient(K, E, case, dtype):
    """gs_conv2d_bwd_weight_bias: gb += sum over pixels of gy from the weight-gradient launches (bf16 MFMA kernels) or the
    channel-sum fallback (fp32 / thin shapes)."""
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    gy = rnd(n, co, h // st, w // st, seed=2).to(dtype).float()
    bw, bb = rnd(ks, ks, ci, co, seed=3), rnd(co, seed=4)
    accw, accb = dev(bw).contiguous(), dev(bb)
    K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), ks, st, 0.2, out=accw, bias_out=accb)
    close(accw, bw + E.conv2d_bwd_weight(x, gy, ks, st, 0.2), rel=1e-4, name="gw +=")
    close(accb, bb + E.channel_sum(gy), rel=1e-4, name="gb +=")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 8, 64, 3, 1), (2, 32, 64, 8, 128, 3, 2), (2, 64, 128, 8, 64, 3, 2),
                                  (4, 256, 256, 2, 16, 3, 1), (2, 64, 32, 6, 40, 3, 1), (2, 2, 32, 8, 64, 1, 1)])
@pytest.mark.parametrize("act", [1, 2])
def test_conv2d_bwd_data_with_activation_mask(K, E, case, dtype, act):
    """gs_conv2d_bwd_data_mask: gx = bwd_data(gy, w) * act'(.) through the activation output that was the conv's input."""
    n, ci, co, h, w, ks, st = case
    wt = rnd(ks, ks, ci, co, seed=2)
    wr = wt.to(dtype).float() if (dtype == torch.bfloat16 and ks == 3) else wt
    gy = rnd(n, co, h // st, w // st, seed=3).to(dtype).float()
    z = E.bias_act_fwd(rnd(n, ci, h, w, seed=4), None, act).to(dtype).float()   # an activation output (both signs for lrelu)
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    ref = E.conv2d_bwd_data(gy, wr, (n, ci, h, w), ks, st, alpha, mask=z, mask_act=act)
    got = K.conv2d_bwd_data(dev(gy, dtype), dev(wt), (n, ci, h, w), ks, st, alpha, mask=dev(z, dtype), mask_act=act)
    close(got, ref, rel=1e-3 if dtype == torch.float32 else 2e-2, name="masked bwd_data")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 32, 8, 64), (4, 256, 2, 16), (8, 512)])
def test_pixel_norm_bwd_with_activation(K, E, shape, dtype):
    x = E.bias_act_fwd(rnd(*shape, seed=1), None, 1).to(dtype).float()
    g = rnd(*shape, seed=2).to(dtype).float()
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    ref = E.pixel_norm_bwd(g, x, 1e-8, act=1)
    close(K.pixel_norm_bwd(dev(g, dtype), dev(x, dtype), 1e-8, act=1), ref, rel=tol, name="pn bwd * lrelu'")
    ad = rnd(*shape, seed=3).to(dtype).float()
    close(K.pixel_norm_bwd(dev(g, dtype), dev(x, dtype), 1e-8, act=1, addend=dev(ad, dtype)), E.pixel_norm_bwd(g, x, 1e-8, act=1, addend=ad), rel=tol,
          name="(pn bwd + addend) * lrelu'")
    close(K.pixel_norm_bwd(dev(g, dtype), dev(x, dtype), 1e-8, pre_act=1), E.pixel_norm_bwd(g, x, 1e-8, pre_act=1), rel=tol, name="pn bwd(g * lrelu')")
    gg = rnd(*shape, seed=4).to(dtype).float()
    close(K.pixel_norm_bwd_bwd(dev(gg, dtype), dev(g, dtype), dev(x, dtype), 1e-8, pre_act=1), E.pixel_norm_bwd_bwd(gg, g, x, 1e-8, pre_act=1),
          rel=1e-3 if dtype == torch.float32 else 3e-2, name="pn bwd_bwd(gg * lrelu')")
    o, og = K.pixel_norm_bwd_bwd(dev(gg, dtype), dev(g, dtype), dev(x, dtype), 1e-8, pre_act=1, with_g=True)
    close(o, E.pixel_norm_bwd_bwd(gg, g, x, 1e-8, pre_act=1), rel=1e-3 if dtype == torch.float32 else 3e-2, name="pair: d/dx")
    close(og, E.pixel_norm_bwd(gg, x, 1e-8, pre_act=1), rel=tol, name="pair: d/dg")


# ---- size-independent properties at the BASELINE layer sizes (the oracle cannot run these shapes in seconds) -----------------
FULL_CASES = [
    (8, 32, 32, 128, 1024, 1),    # top of the pyramid
    (8, 64, 64, 64, 512, 1),
    (8, 32, 64, 128, 1024, 2),    # discriminator downscale conv
    (8, 128, 128, 32, 256, 1),
    (8, 128, 256, 32, 256, 2),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", FULL_CASES)
def test_three_maps_are_mutually_adjoint_at_full_size(K, case, dtype):
    """<conv(x, w), gy> = <x, bwd_data(gy, w)> = <w, bwd_weight(x, gy)> -- the three kernels of a layer are one bilinear form.
    Holds for any correct implementation whatever the size; checked on device in float64 at the BASELINE resolutions."""
    n, ci, co, h, w, st = case
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.randn(n, ci, h, w, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, co, h // st, w // st, device="cuda", generator=g).to(dtype).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(3, 3, ci, co, device="cuda", generator=g)
    if dtype == torch.bfloat16:
        wt = wt.bfloat16().float()   # the conv kernels round the operand copy of the weight; make the three forms see the same values
    alpha = float(np.sqrt(2.0 / (9 * ci)))
    y = K.conv2d_fwd(x, wt, 3, st, alpha)
    gx = K.conv2d_bwd_data(gy, wt, x.shape, 3, st, alpha)
    gw = K.conv2d_bwd_weight(x, gy, 3, st, alpha)
    a = float((y.double() * gy.double()).sum())
    b = float((x.double() * gx.double()).sum())
    c = float((wt.double() * gw.double()).sum())
    scale = float(y.double().norm() * gy.double().norm())   # Cauchy-Schwarz scale of the pairing
    tol = 1e-5 if dtype == torch.float32 else 2e-3          # bf16: y and gx are rounded to 8 bits of mantissa element-wise
    assert abs(a - b) <= tol * scale and abs(a - c) <= tol * scale, (a, b, c, scale)


@pytest.mark.parametrize("case", [(8, 64, 64, 64, 512), (8, 256, 128, 16, 128)])
def test_transposed_conv_is_the_adjoint_of_the_stride2_conv_at_full_size(K, case):
    """conv2d_transpose(x, w) pairs with the stride-2 conv of the flipped roles: <convT(x, w), g> = <x, convT_bwd_data(g, w)> = <w, convT_bwd_weight(x, g)>."""
    n, ci, co, h, w = case
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.randn(n, ci, h, w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    gy = torch.randn(n, co, 2 * h, 2 * w, device="cuda", generator=g).contiguous(memory_format=torch.channels_last)
    wt = torch.randn(3, 3, ci, co, device="cuda", generator=g)
    y = K.conv2d_transpose_fwd(x, wt, 0.05)
    gx = K.conv2d_transpose_bwd_data(gy, wt, 0.05)
    gw = K.conv2d_transpose_bwd_weight(x, gy, 0.05)
    a, b, c = float((y.double() * gy.double()).sum()), float((x.double() * gx.double()).sum()), float((wt.double() * gw.double()).sum())
    scale = float(y.double().norm() * gy.double().norm())
    assert abs(a - b) <= 1e-5 * scale and abs(a - c) <= 1e-5 * scale, (a, b, c, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(1, 64, 64, 6, 40, 3, 1), (1, 64, 128, 12, 72, 3, 2), (3, 128, 64, 5, 33, 3, 1), (1, 32, 32, 1, 7, 3, 1),
                                  (3, 32, 32, 11, 45, 3, 1), (2, 32, 64, 10, 70, 3, 2), (1, 32, 64, 6, 64, 3, 1), (2, 32, 32, 18, 66, 3, 2)])
def test_ragged_and_single_image_shapes(K, E, case, dtype):
    """Tiles that hang over the image edge, odd widths, batch 1 -- through all three maps (64x64-tile weight gradient included; the last four
    cases: the LDS-DMA weight-gradient kernel of the 32-input-channel layers, conv_wgrad_bf16_thin_dma_kernel, in its four instantiations)."""
    n, ci, co, h, w, ks, st = case
    ho, wo = (h + st - 1) // st if st == 1 else h // st, (w + st - 1) // st if st == 1 else w // st
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    wt = rnd(ks, ks, ci, co, seed=2)
    wr = wt.to(dtype).float() if dtype == torch.bfloat16 else wt
    gy = rnd(n, co, ho, wo, seed=3).to(dtype).float()
    alpha = float(np.sqrt(2.0 / (ks * ks * ci)))
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    close(K.conv2d_fwd(dev(x, dtype), dev(wt), ks, st, alpha), E.conv2d_fwd(x, wr, ks, st, alpha), rel=tol, name="fwd")
    close(K.conv2d_bwd_data(dev(gy, dtype), dev(wt), x.shape, ks, st, alpha), E.conv2d_bwd_data(gy, wr, x.shape, ks, st, alpha), rel=tol, name="bwd_data")
    close(K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), ks, st, alpha), E.conv2d_bwd_weight(x, gy, ks, st, alpha), rel=1e-3 if dtype == torch.float32 else 1e-2,
          name="bwd_weight")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pixel_norm_gradient_forms_pair_up_at_full_size(K, dtype):
    """The norm's Jacobian is symmetric, and the pre-/post-activation forms of the fused backward are each other's transpose:
    <gg, (J g) * m> = <g, J (gg * m)> at 8 x 32 x 128 x 1024."""
    g_ = torch.Generator(device="cuda").manual_seed(11)
    shape = (8, 32, 128, 1024)
    z = torch.nn.functional.leaky_relu(torch.randn(*shape, device="cuda", generator=g_), 0.2).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(*shape, device="cuda", generator=g_).to(dtype).contiguous(memory_format=torch.channels_last)
    gg = torch.randn(*shape, device="cuda", generator=g_).to(dtype).contiguous(memory_format=torch.channels_last)
    post = K.pixel_norm_bwd(g, z, 1e-8, act=1)        # (J g) * m
    pre = K.pixel_norm_bwd(gg, z, 1e-8, pre_act=1)    # J (gg * m)
    a, b = float((gg.double() * post.double()).sum()), float((g.double() * pre.double()).sum())
    scale = float(gg.double().norm() * post.double().norm())
    assert abs(a - b) <= (1e-5 if dtype == torch.float32 else 2e-3) * scale, (a, b, scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 32, 128, 1024), (2, 64, 16, 128), (3, 256, 4, 32), (2, 512, 2, 16)])
def test_pixel_norm_backward_sums_its_result_for_the_bias(K, shape, dtype):
    """gs_pixel_norm_bwd_fused_bias: the norm's backward (with the activation derivative and the second gradient folded in) also
    adds sum_pixels of its result into the bias gradient -- the same gradient as the plain call, the same sums as gs_channel_sum
    of it (fp32 accumulation of the unrounded values: equal to bf16 rounding of the summands)."""
    g_ = torch.Generator(device="cuda").manual_seed(11)
    z = torch.nn.functional.leaky_relu(torch.randn(*shape, device="cuda", generator=g_), 0.2).to(dtype).contiguous(memory_format=torch.channels_last)
    g = torch.randn(*shape, device="cuda", generator=g_).to(dtype).contiguous(memory_format=torch.channels_last)
    add = torch.randn(*shape, device="cuda", generator=g_).to(dtype).contiguous(memory_format=torch.channels_last)
    plain = K.pixel_norm_bwd(g, z, 1e-8, act=1, addend=add)
    gb = torch.full((shape[1],), 0.75, device="cuda")
    fused = K.pixel_norm_bwd(g, z, 1e-8, act=1, addend=add, bias_out=gb)
    assert torch.equal(plain, fused)
    ref = plain.double().sum(dim=(0, 2, 3)) + 0.75
    scale = float(plain.double().abs().sum(dim=(0, 2, 3)).max())
    tol = (1e-5 if dtype == torch.float32 else 4e-3) * scale   # (bf16: the reference sums ROUNDED values, the kernel the unrounded ones)
    assert float((gb.double() - ref).abs().max()) <= tol, (float((gb.double() - ref).abs().max()), scale)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_bias_gradient_folds_match_immediate_ones(K, dtype):
    """GS_SUM_PARTIALS + gs_channel_fold_batch: while the trainer defers its parameter gradients, the three producers of a bias
    gradient (activation backward, pixel-norm backward, plain channel sum) only leave their partial rows and ONE launch folds all of
    them at the flush (two when a producer left more than 64 rows) -- the same bits as folding each behind its producer, repeated
    targets accumulate in call order, shapes that are summed directly stay direct."""
    shapes = [(8, 32, 128, 256), (2, 64, 16, 128), (3, 256, 4, 32), (2, 512, 2, 16), (8, 256, 2, 16), (2, 32, 32, 256)]
    g_ = torch.Generator(device="cuda").manual_seed(12)
    work = []
    for shape in shapes + shapes[:2]:
        z = torch.nn.functional.leaky_relu(torch.randn(*shape, device="cuda", generator=g_), 0.2).to(dtype).contiguous(memory_format=torch.channels_last)
        g = torch.randn(*shape, device="cuda", generator=g_).to(dtype).contiguous(memory_format=torch.channels_last)
        work.append((z, g))
    dense = torch.randn(8, 8192, device="cuda", generator=g_).to(dtype)   # few rows, many channels: summed directly, nothing to fold

    def run(deferred):
        tgt = {}
        outs = []
        if deferred:
            K.defer_wgrad_reductions()
        for i, (z, g) in enumerate(work):
            c = z.shape[1]
            for kind in range(3):
                gb = tgt.setdefault((kind, tuple(z.shape)), torch.full((c,), 0.5 + kind, device="cuda"))
                if kind == 0:
                    outs.append(K.act_bwd_bias(g, z, 1, out=gb)[0])
                elif kind == 1:
                    outs.append(K.pixel_norm_bwd(g, z, 1e-8, act=1, bias_out=gb))
                else:
                    K.channel_sum(g, out=gb)
        gd = torch.full((8192,), 0.25, device="cuda")
        K.channel_sum(dense, out=gd)
        if deferred:
            from gansynth_amd import _lib as L
            dt = L.GS_F32 if dtype == torch.float32 else L.GS_BF16
            producer = {0: L.BIAS_FROM_ACT_BWD, 1: L.BIAS_FROM_PIXEL_NORM_BWD, 2: L.BIAS_FROM_CHANNEL_SUM}
            rows = {k: K.lib.gs_bias_partial_rows(producer[k[0]], k[1][0] * k[1][2] * k[1][3], k[1][1], dt) for k in tgt}
            assert len(K._folds) == sum(1 for (z, g) in work for kind in range(3) if rows[(kind, tuple(z.shape))] > 0) >= 2 * len(work)
            assert any(r > 64 for r in rows.values()) and any(0 < r <= 64 for r in rows.values())   # both fold depths
            for k, v in tgt.items():   # untouched until the flush (shapes that are summed directly: already complete)
                assert rows[k] == 0 or float((v - (0.5 + k[0])).abs().max()) == 0.0, k
            K.flush_wgrad_reductions()
            assert K._folds is None
        return tgt, outs, gd

    t0, o0, d0 = run(False)
    t1, o1, d1 = run(True)
    assert torch.equal(d0, d1)
    for a, b in zip(o0, o1):
        assert torch.equal(a, b)
    for k in t0:
        assert torch.equal(t0[k], t1[k]), k
    z, g = work[0]
    ref = g.double().sum(dim=(0, 2, 3)) + work[len(shapes)][1].double().sum(dim=(0, 2, 3)) + 2.5   # the repeated shape: two sums on top of the initial 2.5
    assert float((t1[(2, tuple(z.shape))].double() - ref).abs().max()) <= 1e-4 * float(g.double().abs().sum(dim=(0, 2, 3)).max())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_weight_gradient_reductions_match_immediate_ones(K, dtype):
    """Deferred weight gradients (gs_*_bwd_weight*_multi + gs_wgrad_reduce_batch): the (x, gy) pairs of a layer contracted by one
    launch and the slice partials of all layers folded together -- equal (to fp32 rounding) to one immediate call per pair,
    including sources without a bias contribution, the transposed conv, the thin / direct kernels and more layers than one
    reduction launch holds."""
    cases = [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 8, 64, 3, 1), (2, 64, 128, 8, 64, 3, 2), (2, 32, 64, 8, 128, 3, 2), (2, 128, 128, 4, 32, 3, 1),
             (2, 32, 2, 8, 64, 1, 1), (2, 2, 32, 8, 64, 1, 1), (4, 1, 16, 2, 16, 3, 1), (2, 256, 256, 2, 16, 3, 1)]
    cases = cases + cases[:5] + cases[:5]   # > 16 entries, repeated targets
    work = []
    for i, (n, ci, co, h, w, ks, st) in enumerate(cases):
        x = dev(rnd(n, ci, h, w, seed=10 + i), dtype)
        gy = dev(rnd(n, co, h // st, w // st, seed=40 + i), dtype)
        work.append((x, gy, ks, st))
    xt, gyt = dev(rnd(2, 64, 8, 64, seed=4), dtype), dev(rnd(2, 32, 16, 128, seed=5), dtype)

    def run(deferred):
        grads = {}
        for (x, gy, ks, st) in work:   # repeated cases share their gradient buffers
            key = (tuple(x.shape), tuple(gy.shape), ks, st)
            if key not in grads:
                grads[key] = (torch.full((ks, ks, x.shape[1], gy.shape[1]), 0.5, device="cuda"), torch.full((gy.shape[1],), -0.25, device="cuda"))
        gt = torch.full((3, 3, 64, 32), 0.125, device="cuda")
        if deferred:
            K.defer_wgrad_reductions()
        for i, (x, gy, ks, st) in enumerate(work):
            gw, gb = grads[(tuple(x.shape), tuple(gy.shape), ks, st)]
            # (fp32 bias sums are not deferrable; the third round contributes to the weights only, like a second-order term)
            K.conv2d_bwd_weight(x, gy, ks, st, 0.2, out=gw, bias_out=gb if (ks == 3 and dtype == torch.bfloat16 and i < 14) else None)
        K.conv2d_transpose_bwd_weight(xt, gyt, 0.1, out=gt)
        K.conv2d_transpose_bwd_weight(xt, gyt, 0.3, out=gt)
        if deferred:
            assert K.flush_wgrad_reductions() > 16
        torch.cuda.synchronize()
        return [t.clone() for pair in grads.values() for t in pair] + [gt.clone()]

    now, later = run(False), run(True)
    for a, b in zip(now, later):   # (the batched kernel sums the slices in a different association: equal to rounding, not to the bit)
        assert float((a - b).abs().max()) <= 1e-5 * float(a.abs().max()), (a.shape, float((a - b).abs().max()), float(a.abs().max()))
    assert K._pending is None


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_multi_source_weight_gradient_matches_oracle(K, E, dtype):
    """gs_conv2d_bwd_weight_bias_multi against the CPU restatement: gw = sum over the pairs, gb over the masked ones."""
    for (n, ci, co, h, w, ks, st) in [(2, 64, 64, 8, 64, 3, 1), (2, 32, 64, 8, 128, 3, 2), (1, 128, 64, 6, 40, 3, 1)]:
        xs = [rnd(n, ci, h, w, seed=70 + i).to(dtype).float() for i in range(3)]
        gys = [rnd(n, co, h // st, w // st, seed=80 + i).to(dtype).float() for i in range(3)]
        gw = torch.zeros(ks, ks, ci, co, device="cuda")
        gb = torch.zeros(co, device="cuda")
        with_b = dtype == torch.bfloat16
        K.defer_wgrad_reductions()
        for i in range(3):
            K.conv2d_bwd_weight(dev(xs[i], dtype), dev(gys[i], dtype), ks, st, 0.3, out=gw, bias_out=gb if (with_b and i != 1) else None)
        K.flush_wgrad_reductions()
        ref = sum(E.conv2d_bwd_weight(xs[i], gys[i], ks, st, 0.3) for i in range(3))
        close(gw, ref, rel=1e-4, name=f"multi wgrad {ci}->{co} s{st}")
        if with_b:
            close(gb, E.channel_sum(gys[0]) + E.channel_sum(gys[2]), rel=1e-4, name="multi wgrad bias (sources 0 and 2)")


def test_grouped_weight_gradients_match_oracle(K, E):
    """gs_conv_wgrad_jobs: the >= 64-channel bf16 layers of a backward pass run as stream-K groups (one launch per conv mode and
    tile width over the pixel tiles of ALL their layers, blocks crossing layer and channel-tile boundaries) -- every gradient
    against the CPU restatement: layers of different sizes in one group, several sources with different image counts and bias
    masks, the transposed conv (swapped sides, transposed store), 16-wide tiles, more layers than a group holds and a layer
    whose sources come in two jobs adding into one gradient."""
    dtype = torch.bfloat16
    layers = [("conv", [3, 2], 64, 64, 8, 64, 1), ("conv", [2], 128, 64, 16, 96, 1), ("conv", [1, 1, 2], 64, 128, 6, 40, 1), ("conv", [2], 256, 256, 4, 32, 1),
              ("conv", [2, 2], 64, 128, 16, 64, 2), ("conv", [3], 128, 128, 8, 128, 2), ("conv", [2], 64, 64, 2, 16, 1), ("conv", [2, 1], 128, 64, 4, 16, 1),
              ("conv", [2], 64, 64, 4, 32, 2), ("convT", [2, 1], 128, 64, 8, 32, 2), ("convT", [2], 64, 64, 2, 16, 2),
              ("conv", [1, 1, 1, 1, 1, 1], 64, 64, 8, 32, 1)]
    layers = layers + [("conv", [1], 64, 64, 8, 32 + 32 * i, 1) for i in range(18)]   # one mode / width: more than GS_SK_MAX_JOBS layers
    K.defer_wgrad_reductions()
    want, got = [], []
    for li, (kind, ns, ci, co, h, w, st) in enumerate(layers):
        gw = torch.full((3, 3, ci, co), 0.25, device="cuda")
        gb = torch.full((co,), -0.5, device="cuda") if kind == "conv" else None
        ref_w, ref_b = torch.full((3, 3, ci, co), 0.25), (torch.full((co,), -0.5) if kind == "conv" else None)
        for si, n in enumerate(ns):
            x = rnd(n, ci, h, w, seed=100 * li + si).to(dtype).float()
            if kind == "conv":
                gy = rnd(n, co, h // st, w // st, seed=100 * li + 50 + si).to(dtype).float()
                with_b = si != 1
                K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), 3, st, 0.3, out=gw, bias_out=gb if with_b else None)
                ref_w += E.conv2d_bwd_weight(x, gy, 3, st, 0.3)
                if with_b:
                    ref_b += E.channel_sum(gy)
            else:
                gy = rnd(n, co, 2 * h, 2 * w, seed=100 * li + 50 + si).to(dtype).float()
                K.conv2d_transpose_bwd_weight(dev(x, dtype), dev(gy, dtype), 0.3, out=gw)
                ref_w += E.conv2d_transpose_bwd_weight(x, gy, 0.3)
        want.append((ref_w, ref_b))
        got.append((gw, gb))
    # the 257-input-channel conv evaluated on two channel slices of one variable (networks.py:174-176): both slice gradients are
    # added straight into their rows of the variable's gradient (a strided target), the 64-channel slice by the grouped kernel,
    # the 1-channel one by the direct kernel's pending reduction -- two sources each
    parent = torch.full((3, 3, 65, 64), 0.125, device="cuda")
    ref_parent = torch.full((3, 3, 65, 64), 0.125)
    extra = 0
    slices = () if os.environ.get("GS_NO_WGRAD_GROUPS") else ((0, 64), (64, 65))   # (the measurement knob also turns slice targets off)
    for lo, hi in slices:
        assert K.wgrad_slice_target_ok(torch.empty(1, hi - lo, 1, 1, dtype=dtype), 64, 3, 1)
        for si in range(2):
            x = rnd(2, hi - lo, 4, 32, seed=900 + 10 * lo + si).to(dtype).float()
            gy = rnd(2, 64, 4, 32, seed=950 + 10 * lo + si).to(dtype).float()
            K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), 3, 1, 0.3, out=parent[:, :, lo:hi, :])
            ref_parent[:, :, lo:hi, :] += E.conv2d_bwd_weight(x, gy, 3, 1, 0.3)
            extra += 1
    assert K.flush_wgrad_reductions() == sum(len(l[1]) for l in layers) + extra
    close(parent, ref_parent, rel=1e-4, name="channel-slice targets of one variable")
    for li, ((rw, rb), (gw, gb)) in enumerate(zip(want, got)):
        close(gw, rw, rel=1e-4, name=f"grouped wgrad layer {li} {layers[li]}")
        if rb is not None:
            close(gb, rb, rel=1e-4, name=f"grouped wgrad bias layer {li}")


def test_large_layers_contracted_early_and_on_fewer_cus(K, E):
    """kernels.early_flush_rule + flush_wgrad_reductions(select=) + gs_wgrad_cu_cap (what models.GANSynth._early_flush does on the forked branch
    of a run's hipGraph): a backward pass records the pairs of large layers, then a small layer -- the rule fires ONCE, at that point, with a
    selector for the large ones; they are contracted then (launches sized for 192 or 64 of the CUs), a second pair of one of them and everything
    small at the final flush.  Every gradient against the CPU restatement; the cap returns its previous setting and 0 restores the whole chip."""
    dtype = torch.bfloat16
    # (kind, images per pair, ci, co, h, w, stride): two 32-input-channel layers (the LDS-DMA kernel), two stream-K group layers, the 1x1 colour conv; then small ones
    large = [("conv", [8], 32, 32, 64, 256, 1), ("conv", [4, 4], 32, 64, 64, 256, 2), ("conv", [8], 64, 64, 32, 128, 1), ("conv", [4], 64, 128, 32, 128, 2), ("conv1", [8], 2, 32, 64, 256, 1)]
    small = [("conv", [8], 128, 128, 4, 32, 1), ("conv", [8], 256, 256, 2, 16, 1)]
    for cap in (192, 64):
        fired, flushed = [], []
        K.defer_wgrad_reductions()

        def early(select):
            fired.append(sorted(K._key_pixels(k) for k in K._pending))
            was = K.lib.gs_wgrad_cu_cap(cap)
            assert was == 0
            flushed.append(K.flush_wgrad_reductions(select=select))
            assert K.lib.gs_wgrad_cu_cap(0) == cap
        K.early_flush_rule(32 * 128 // 4, early)
        want, got = [], []

        def record(li, kind, ns, ci, co, h, w, st, first=0):
            ks = 1 if kind == "conv1" else 3
            if first == 0:
                got.append((torch.full((ks, ks, ci, co), 0.25, device="cuda"), torch.full((co,), -0.5, device="cuda")))
                want.append([torch.full((ks, ks, ci, co), 0.25), torch.full((co,), -0.5)])
            gw, gb = got[li]
            for si, n in enumerate(ns):
                x = rnd(n, ci, h, w, seed=1000 * cap + 100 * li + si + 7 * first).to(dtype).float()
                gy = rnd(n, co, h // st, w // st, seed=1000 * cap + 100 * li + 50 + si + 7 * first).to(dtype).float()
                K.conv2d_bwd_weight(dev(x, dtype), dev(gy, dtype), ks, st, 0.3, out=gw, bias_out=gb if ks == 3 else None)
                want[li][0] += E.conv2d_bwd_weight(x, gy, ks, st, 0.3)
                if ks == 3:
                    want[li][1] += E.channel_sum(gy)
        for li, l in enumerate(large):
            record(li, *l)
        assert not fired
        for li, l in enumerate(small):
            record(len(large) + li, *l)
        assert len(fired) == 1 and flushed == [sum(len(l[1]) for l in large)], (fired, flushed)   # once, at the first small layer: every large pair so far
        record(1, *large[1], first=1)                     # a late pair of a layer that was already contracted: adds into the same gradient at the end
        K.early_flush_rule(0, None)
        assert K.flush_wgrad_reductions() == sum(len(l[1]) for l in small) + len(large[1][1])
        for li, ((gw, gb), (rw, rb)) in enumerate(zip(got, want)):
            close(gw, rw, rel=1e-4, name=f"cap {cap}: layer {li}")
            if gw.shape[0] == 3:
                close(gb, rb, rel=1e-4, name=f"cap {cap}: bias {li}")


def test_throwaway_streams(K):
    """gs_streams_create / gs_streams_destroy (models.GANSynth._leveled_queues: the HIP runtime's pool of hardware queues is level while a graph
    with parallel branches is instantiated): n distinct streams that take work, destroyed without error; n = 0 is fine, n < 0 is refused."""
    import ctypes
    n = 12
    h = (ctypes.c_void_p * n)()
    p = ctypes.cast(h, ctypes.POINTER(ctypes.c_void_p))
    assert K.lib.gs_streams_create(n, p) == 0
    assert len({int(v) for v in h}) == n and all(v for v in h)
    x = torch.ones(1024, device="cuda")
    with torch.cuda.stream(torch.cuda.ExternalStream(int(h[3]))):
        y = x * 2
    torch.cuda.synchronize()
    assert float(y.sum()) == 2048.0
    assert K.lib.gs_streams_destroy(n, p) == 0 and all(not v for v in h)
    assert K.lib.gs_streams_create(0, p) == 0
    assert K.lib.gs_streams_create(-1, p) != 0


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [("conv", 2, 32, 32, 8, 128), ("conv", 8, 64, 64, 64, 512), ("conv", 2, 64, 64, 8, 64), ("conv", 2, 256, 256, 4, 32),
                                  ("convT", 2, 64, 32, 8, 64), ("convT", 8, 128, 64, 32, 256), ("convT", 2, 256, 256, 4, 32)])
def test_conv_bias_act_norm_in_one_call(K, E, case, dtype):
    """gs_conv2d[_transpose_s2]_fwd_bias_act_norm: (z, y) = (act(conv + b), pixel_norm(z)) -- fused into the conv epilogue for the
    32- / 64-channel tiles (first, second, fifth and sixth case), conv + separate norm pass otherwise; with and without z."""
    kind, n, ci, co, h, w = case
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    wt = rnd(3, 3, ci, co, seed=2)
    bias = rnd(co, seed=3, scale=0.1)
    eps, alpha = 1e-8, 0.05
    if kind == "conv":
        zr = E.conv2d_fwd_bias_act(x, wt, bias, 3, 1, alpha, 1)
        run = lambda want_z: K.conv2d_fwd_bias_act_norm(dev(x, dtype), dev(wt), dev(bias), 3, 1, alpha, 1, eps, want_z=want_z)
    else:
        zr = E.conv2d_transpose_fwd_bias_act(x, wt, bias, alpha, 1)
        run = lambda want_z: K.conv2d_transpose_fwd_bias_act_norm(dev(x, dtype), dev(wt), dev(bias), alpha, 1, eps, want_z=want_z)
    yr = E.pixel_norm_fwd(zr, eps)
    tol = 1e-3 if dtype == torch.float32 else 2e-2
    z, y = run(True)
    close(z, zr, rel=tol, name=f"{case} z")
    close(y, yr, rel=tol, name=f"{case} y")
    z2, y2 = run(False)
    assert z2 is None
    close(y2, yr, rel=tol, name=f"{case} y (no z)")
    assert torch.equal(y, y2) or dtype == torch.bfloat16   # fp32: the same fused arithmetic either way


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(2, 32, 32, 8, 128, 3, 1), (2, 64, 64, 8, 64, 3, 1), (2, 64, 128, 8, 64, 3, 2), (2, 32, 64, 8, 128, 3, 2), (2, 2, 32, 8, 64, 1, 1)])
def test_conv2d_fwd_with_activation_mask(K, E, case, dtype):
    """gs_conv2d_fwd_mask: conv2d(x, w) * lrelu'(.) through an activation output of the result's shape (epilogue for >= 64 output
    channels, in place after the conv otherwise)."""
    n, ci, co, h, w, ks, st = case
    x = rnd(n, ci, h, w, seed=1).to(dtype).float()
    wt = rnd(ks, ks, ci, co, seed=2)
    z = E.bias_act_fwd(rnd(n, co, h // st, w // st, seed=3), None, 1).to(dtype).float()
    ref = E.act_bwd(E.conv2d_fwd(x, wt, ks, st, 0.1), z, 1)
    got = K.conv2d_fwd_mask(dev(x, dtype), dev(wt), ks, st, 0.1, dev(z, dtype), 1)
    close(got, ref, rel=1e-3 if dtype == torch.float32 else 2e-2, name=f"fwd mask {case}")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gan_losses_in_one_launch(K, dtype):
    """gs_gan_d_loss / gs_gan_g_loss against the loss algebra of models.py:39-65 written out in torch (value and gradients)."""
    import torch.nn.functional as TF
    n, c = 8, 61
    g = torch.Generator().manual_seed(3)
    real = (torch.randn(n, c, generator=g) * 3).to(dtype).float().requires_grad_(True)
    fake = (torch.randn(n, c, generator=g) * 3).to(dtype).float().requires_grad_(True)
    lab = torch.nn.functional.one_hot(torch.randint(0, c, (n,), generator=g), c).float()
    pen = (torch.rand(n, generator=g) * 2).requires_grad_(True)
    ssq = (torch.rand(n, generator=g) * 1e-3).requires_grad_(True)
    ld = (TF.softplus(-(real * lab).sum(1)) + TF.softplus((fake * lab).sum(1)) + 5.0 * pen).mean()
    g_real, g_fake, g_pen = torch.autograd.grad(ld, [real, fake, pen])
    loss, kr, kf, kp = K.gan_d_loss(dev(real.detach(), dtype), dev(fake.detach(), dtype), dev(lab, dtype), pen.detach().cuda(), 5.0)
    close(kp, g_pen, rel=1e-6, name="d loss: d/d penalty")
    assert K.gan_d_loss(dev(real.detach(), dtype), dev(fake.detach(), dtype), dev(lab, dtype), None)[3] is None
    tol = 1e-5 if dtype == torch.float32 else 1e-2
    assert abs(float(loss) - float(ld.detach())) <= 1e-5 * abs(float(ld.detach()))
    close(kr, g_real, rel=tol, name="d loss: d/d real logits")
    close(kf, g_fake, rel=tol, name="d loss: d/d fake logits")
    assert torch.allclose(g_pen, torch.full((n,), 5.0 / n))
    lg = (TF.softplus(-(fake * lab).sum(1)) + 0.1 / (ssq + 1e-6)).mean()
    g_fake2, g_ssq = torch.autograd.grad(lg, [fake, ssq])
    loss, kf, ks = K.gan_g_loss(dev(fake.detach(), dtype), dev(lab, dtype), ssq.detach().cuda(), 0.1, 1e-6)
    assert abs(float(loss) - float(lg.detach())) <= 1e-5 * abs(float(lg.detach()))
    close(kf, g_fake2, rel=tol, name="g loss: d/d fake logits")
    close(ks, g_ssq, rel=1e-5, name="g loss: d/d sumsq")
    loss, kf, ks = K.gan_g_loss(dev(fake.detach(), dtype), dev(lab, dtype), None, 0.0, 1e-6)
    assert ks is None and abs(float(loss) - float(TF.softplus(-(fake.detach() * lab).sum(1)).mean())) <= 1e-5
    # either half of a sum as a launch of its own (two passes on two streams: include/gansynth_hip.h): the same gradients bit for bit, and the
    # two partial means add up to the loss
    _, kr0, kf0, kp0 = K.gan_d_loss(dev(real.detach(), dtype), dev(fake.detach(), dtype), dev(lab, dtype), pen.detach().cuda(), 5.0)
    l_real, kr1, none_f, kp1 = K.gan_d_loss(dev(real.detach(), dtype), None, dev(lab, dtype), pen.detach().cuda(), 5.0)
    l_fake, none_r, kf1, none_p = K.gan_d_loss(None, dev(fake.detach(), dtype), dev(lab, dtype), None, 1.0)
    assert none_f is None and none_r is None and none_p is None
    assert torch.equal(kr0, kr1) and torch.equal(kf0, kf1) and torch.equal(kp0, kp1)
    assert abs(float(l_real) + float(l_fake) - float(ld.detach())) <= 1e-5 * abs(float(ld.detach()))
    _, kf0, ks0 = K.gan_g_loss(dev(fake.detach(), dtype), dev(lab, dtype), ssq.detach().cuda(), 0.1, 1e-6)
    l_ms, none_f, ks1 = K.gan_g_loss(None, None, ssq.detach().cuda(), 0.1, 1e-6)
    l_adv, kf1, none_s = K.gan_g_loss(dev(fake.detach(), dtype), dev(lab, dtype), None, 0.0, 1e-6)
    assert none_f is None and none_s is None and torch.equal(kf0, kf1) and torch.equal(ks0, ks1)
    assert abs(float(l_ms) + float(l_adv) - float(lg.detach())) <= 1e-5 * abs(float(lg.detach()))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [("conv", 8, 32, 32, 128, 1024), ("conv", 8, 64, 64, 64, 512), ("convT", 8, 64, 32, 64, 512),   # the three fused shapes
                                  ("conv", 2, 32, 32, 8, 128), ("conv", 2, 64, 64, 8, 64), ("convT", 2, 64, 32, 8, 64), ("conv", 2, 256, 256, 4, 32),
                                  ("convT", 2, 128, 64, 4, 32), ("conv", 1, 64, 64, 6, 40),
                                  ("conv1", 8, 32, 2, 128, 1024), ("conv1", 2, 64, 2, 8, 64), ("conv1", 2, 256, 2, 4, 32), ("conv1", 3, 128, 2, 5, 7)])   # the colour block (1x1)
@pytest.mark.parametrize("with_addend", [False, True])
def test_data_gradient_continued_through_the_previous_pixel_norm(K, E, case, dtype, with_addend):
    """gs_conv2d[_transpose_s2]_bwd_data_pnbwd: (pixel_norm_bwd(B^T(gy, w), z) + addend) * leaky_relu'(z) in the conv's epilogue (the 32- /
    64-channel full-size layers) or as conv + in-place norm backward (every other shape).  Reference = the float64 definition of the norm
    backward applied to the ORACLE-side data gradient (tests/cpu_kernels.py: torch-CPU autograd of oracle.torch_ref's conv on the same
    inputs -- nothing of the HIP path feeds the reference); second, the two separate HIP kernels.  In bf16 the separate path rounds the
    intermediate gradient to bf16, the fused one does not: compared at bf16 resolution of the tensor's scale; fp32 at 1e-5."""
    kind, n, ci, co, h, w = case   # ci: channels of z / gx (the conv's input side), co: of gy
    gen = torch.Generator(device="cuda").manual_seed(11)
    CL = torch.channels_last
    z = torch.nn.functional.leaky_relu(torch.randn(n, ci, h, w, device="cuda", generator=gen), 0.2).to(dtype).contiguous(memory_format=CL)
    ks = 1 if kind == "conv1" else 3
    wt = torch.randn(ks, ks, ci, co, device="cuda", generator=gen)
    oh, ow = (2 * h, 2 * w) if kind == "convT" else (h, w)
    gy = torch.randn(n, co, oh, ow, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
    add = (0.5 * torch.randn(n, ci, h, w, device="cuda", generator=gen)).to(dtype).contiguous(memory_format=CL) if with_addend else None
    alpha, eps, act = 0.05, 1e-8, 1
    if kind != "convT":
        g = K.conv2d_bwd_data(gy, wt, (n, ci, h, w), ks, 1, alpha)
        got = K.conv2d_bwd_data_pnbwd(gy, wt, (n, ci, h, w), ks, 1, alpha, z, eps, act, addend=add)
    else:
        g = K.conv2d_transpose_bwd_data(gy, wt, alpha)
        got = K.conv2d_transpose_bwd_data_pnbwd(gy, wt, alpha, z, eps, act, addend=add)
    ref = K.pixel_norm_bwd(g, z, eps, act=act, addend=add)
    # float64 evaluation of the definition on the oracle-side fp32 data gradient (independent of the HIP conv and of the norm kernels)
    gg = (E.conv2d_bwd_data(gy.float().cpu(), wt.cpu(), (n, ci, h, w), ks, 1, alpha) if kind != "convT" else
          E.conv2d_transpose_bwd_data(gy.float().cpu(), wt.cpu(), alpha)).cuda().double()
    zz = z.double()
    r = torch.rsqrt((zz * zz).mean(dim=1, keepdim=True) + eps)
    want = r * (gg - zz * r * r * (zz * gg).mean(dim=1, keepdim=True))
    if add is not None:
        want = want + add.double()
    want = want * torch.where(zz > 0, 1.0, 0.2)
    scale = float(want.abs().max())
    tol = 1e-5 if dtype == torch.float32 else 2e-2
    assert float((got.double() - want).abs().max()) <= tol * scale, (float((got.double() - want).abs().max()), scale)
    assert float((got.double() - ref.double()).abs().max()) <= (1e-5 if dtype == torch.float32 else 3e-2) * scale
    if dtype == torch.bfloat16:   # one rounding instead of two: at least as close to the definition as the two-kernel path
        assert float((got.double() - want).pow(2).mean()) <= 1.05 * float((ref.double() - want).pow(2).mean()) + 1e-12



@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [("conv", 8, 32, 32, 128, 1024), ("conv", 8, 64, 64, 64, 512), ("convT", 8, 64, 32, 64, 512),   # fused at full size
                                  ("convT", 8, 128, 64, 32, 256), ("conv", 2, 32, 32, 8, 128), ("conv", 2, 8, 64, 8, 64), ("convT", 2, 16, 32, 8, 64),
                                  ("conv", 2, 256, 256, 4, 32), ("convT", 2, 128, 64, 4, 32), ("conv", 1, 40, 64, 6, 40), ("conv", 3, 5, 8, 5, 7)])
def test_second_order_norm_gradients_in_the_forward_conv(K, E, case, dtype):
    """gs_conv2d[_transpose_s2]_fwd_pnbwdbwd: with t = conv(x, w) the cotangent of u = act'(z) pixel_norm_bwd(g, z), both gradients of that
    node (w.r.t. g and w.r.t. z) from the conv's epilogue, against a float64 autograd evaluation of the definition on the ORACLE-side fp32
    conv output (tests/cpu_kernels.py: oracle.torch_ref's conv on the same inputs -- nothing of the HIP path feeds the reference) and against
    the two-kernel path (conv, then gs_pixel_norm_bwd_bwd_fused).  bf16: compared at bf16 resolution of each tensor's scale."""
    kind, n, ci, co, h, w = case
    gen = torch.Generator(device="cuda").manual_seed(5)
    CL = torch.channels_last
    x = torch.randn(n, ci, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
    wt = torch.randn(3, 3, ci, co, device="cuda", generator=gen)
    oh, ow = (2 * h, 2 * w) if kind == "convT" else (h, w)
    z = torch.nn.functional.leaky_relu(torch.randn(n, co, oh, ow, device="cuda", generator=gen), 0.2).to(dtype).contiguous(memory_format=CL)
    g = torch.randn(n, co, oh, ow, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
    alpha, eps, act = 0.03, 1e-8, 1
    if kind == "conv":
        t = K.conv2d_fwd(x, wt, 3, 1, alpha)
        t32 = E.conv2d_fwd(x.float().cpu(), wt.cpu(), 3, 1, alpha).cuda()
        got_z, got_g = K.conv2d_fwd_pnbwdbwd(x, wt, 3, 1, alpha, g, z, eps, act)
    else:
        t = K.conv2d_transpose_fwd(x, wt, alpha)
        t32 = E.conv2d_transpose_fwd(x.float().cpu(), wt.cpu(), alpha).cuda()
        got_z, got_g = K.conv2d_transpose_fwd_pnbwdbwd(x, wt, alpha, g, z, eps, act)
    ref_z, ref_g = K.pixel_norm_bwd_bwd(t, g, z, eps, pre_act=act, with_g=True)
    zz, gg = z.double().requires_grad_(True), g.double().requires_grad_(True)
    r = torch.rsqrt((zz * zz).mean(dim=1, keepdim=True) + eps)
    u = r * (gg - zz * r * r * (zz * gg).mean(dim=1, keepdim=True)) * torch.where(zz > 0, 1.0, 0.2)
    want_g, want_z = torch.autograd.grad((u * t32.double()).sum(), [gg, zz])
    for name, got, ref, want in (("g", got_g, ref_g, want_g), ("z", got_z, ref_z, want_z)):
        scale = float(want.abs().max())
        err = float((got.double() - want).abs().max())
        assert err <= (2e-5 if dtype == torch.float32 else 2e-2) * scale, (name, err, scale)
        assert float((got.double() - ref.double()).abs().max()) <= (2e-5 if dtype == torch.float32 else 3e-2) * scale, name
        if dtype == torch.bfloat16:
            assert float((got.double() - want).pow(2).mean()) <= 1.05 * float((ref.double() - want).pow(2).mean()) + 1e-12, name


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 256, 2, 16), (3, 5, 2, 3), (4, 64, 4, 32)])
def test_dense_units_to_channels_last_in_the_activation_pass(K, shape, dtype):
    """gs_units_bias_act_to_nhwc / gs_nhwc_act_bwd_to_units (networks.py:51-55: tf.reshape of the dense layer's channel-major units to
    [n, c, h, w], leaky_relu) against the three separate steps written in torch: exact (a reorder, one add, one select)."""
    n, c, h, w = shape
    gen = torch.Generator(device="cuda").manual_seed(2)
    y = torch.randn(n, c * h * w, device="cuda", generator=gen).to(dtype)
    bias = torch.randn(c * h * w, device="cuda", generator=gen)
    want = torch.nn.functional.leaky_relu((y.float() + bias).reshape(n, c, h, w), 0.2).to(dtype)
    z = K.units_bias_act_to_nhwc(y, bias, c, h, w, 1)
    assert z.is_contiguous(memory_format=torch.channels_last) and torch.equal(z, want)
    assert torch.equal(K.units_bias_act_to_nhwc(y, None, c, h, w, 0), y.reshape(n, c, h, w))
    g = torch.randn(n, c, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=torch.channels_last)
    gu = K.nhwc_act_bwd_to_units(g, z, 1)
    assert torch.equal(gu, (g.float() * torch.where(z > 0, 1.0, 0.2)).to(dtype).reshape(n, c * h * w))
    ggu = torch.randn(n, c * h * w, device="cuda", generator=gen).to(dtype)
    gg = K.units_bias_act_to_nhwc(ggu, None, c, h, w, 1, mask=z)   # the adjoint map (second-order pass)
    assert torch.equal(gg, (ggu.float().reshape(n, c, h, w) * torch.where(z > 0, 1.0, 0.2)).to(dtype))
    # <gu, ggu> == <g, gg>: the two maps are adjoint
    a, b = float((gu.double() * ggu.double()).sum()), float((g.double() * gg.double()).sum())
    assert abs(a - b) <= (1e-6 if dtype == torch.float32 else 2e-2) * max(1.0, abs(a))


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(8, 2, 32, 128, 1024), (2, 2, 64, 16, 128), (3, 1, 32, 5, 7), (2, 3, 8, 4, 4)])
def test_colour_block_forward_with_the_next_node_s_mask(K, shape, dtype):
    """gs_conv2d_fwd_mask on the 1x1 colour -> features conv (second-order pass of the R1 term): the streaming kernel multiplies by
    leaky_relu'(.) through the mask itself -- same values as the conv followed by gs_act_bwd (one rounding instead of two in bf16)."""
    n, ci, co, h, w = shape
    gen = torch.Generator(device="cuda").manual_seed(6)
    CL = torch.channels_last
    x = torch.randn(n, ci, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
    wt = torch.randn(1, 1, ci, co, device="cuda", generator=gen)
    mask = torch.randn(n, co, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
    got = K.conv2d_fwd_mask(x, wt, 1, 1, 0.7, mask, 1)
    want = (K.conv2d_fwd(x.float(), wt, 1, 1, 0.7) * torch.where(mask.float() > 0, 1.0, 0.2)).cpu()
    close(got, want, rel=1e-6 if dtype == torch.float32 else 1e-2, name="colour block fwd mask")


def _sign_words(z):
    """The 1-bit mask layout of include/gansynth_hip.h restated with torch: one 32-bit word per (pixel, 32-channel tile), bit 8 (2 h + q) + k =
    channel 16 q + 8 h + k of the tile is > 0."""
    n, c, h, w = z.shape
    pos = (z.permute(0, 2, 3, 1).float() > 0).reshape(-1, c // 32, 2, 2, 8).to(torch.int64)   # [pixel, tile, q, h, k]
    q, hh, k = torch.meshgrid(torch.arange(2), torch.arange(2), torch.arange(8), indexing="ij")
    shift = (8 * (2 * hh + q) + k).to(z.device)
    word = (pos << shift).sum(dim=(2, 3, 4))
    return torch.where(word >= 2 ** 31, word - 2 ** 32, word).to(torch.int32).reshape(-1)


def _stored_words(z):
    return torch.empty(0, dtype=torch.int32, device=z.device).set_(z.untyped_storage(), z.numel() // 2, (z.numel() // 32,))


@pytest.mark.parametrize("case", [(2, 32, 32, 16, 128, 3, 1), (2, 64, 64, 8, 64, 3, 1), (2, 32, 64, 8, 128, 3, 2), (3, 128, 128, 8, 32, 3, 1), (2, 256, 256, 4, 16, 3, 1),
                                  (2, 2, 32, 16, 128, 1, 1), (3, 2, 64, 8, 72, 1, 1), (2, 32, 96, 6, 40, 3, 1)])
def test_one_bit_leaky_relu_masks(K, case):
    """GS_ACT_WRITE_BITS / GS_ACT_LRELU_BITS: a bf16 leaky-relu conv result carries its sign bits behind it (written by the MFMA epilogue, or by
    gs_pack_act_bits after the direct kernels), and the masked launches that read them give bit-identical results to the ones that read the
    values.  An activation without the bits (a clone) takes the values path."""
    from gansynth_amd import kernels
    n, ci, co, h, w, ks, st = case
    gen = torch.Generator(device="cuda").manual_seed(11)
    CL = torch.channels_last
    x = torch.randn(n, ci, h, w, device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
    wt = torch.randn(ks, ks, ci, co, device="cuda", generator=gen)
    bias = torch.randn(co, device="cuda", generator=gen)
    z = K.conv2d_fwd_bias_act(x, wt, bias, ks, st, 0.1, 1)
    plain = K.conv2d_fwd_bias_act(x, wt, bias, ks, st, 0.1, 1, bits=False)
    assert kernels._has_bits(z) and not kernels._has_bits(plain)
    assert torch.equal(z, plain)
    assert torch.equal(_stored_words(z), _sign_words(z)), "sign words"
    zc = z.clone(memory_format=CL)
    assert not kernels._has_bits(zc)
    # z as the mask of a data gradient (the conv whose INPUT z was) and of a forward-on-cotangents conv (whose output has z's shape)
    for c2 in (co, 64):
        w2 = torch.randn(3, 3, co, c2, device="cuda", generator=gen)
        gy = torch.randn(n, c2, z.shape[2], z.shape[3], device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
        a = K.conv2d_bwd_data(gy, w2, tuple(z.shape), 3, 1, 0.07, mask=z, mask_act=1)
        b = K.conv2d_bwd_data(gy, w2, tuple(z.shape), 3, 1, 0.07, mask=zc, mask_act=1)
        assert torch.equal(a, b), f"bwd_data mask, {c2} channels"
        if z.shape[2] % 2 == 0:   # behind a stride-2 conv: the transposed-conv-shaped data gradient
            gy2 = torch.randn(n, c2, z.shape[2] // 2, z.shape[3] // 2, device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
            a = K.conv2d_bwd_data(gy2, w2, tuple(z.shape), 3, 2, 0.07, mask=z, mask_act=1)
            b = K.conv2d_bwd_data(gy2, w2, tuple(z.shape), 3, 2, 0.07, mask=zc, mask_act=1)
            assert torch.equal(a, b), f"stride-2 bwd_data mask, {c2} channels"
        w3 = torch.randn(3, 3, c2, co, device="cuda", generator=gen)
        xx = torch.randn(n, c2, z.shape[2], z.shape[3], device="cuda", generator=gen).bfloat16().contiguous(memory_format=CL)
        a = K.conv2d_fwd_mask(xx, w3, 3, 1, 0.07, z, 1)
        b = K.conv2d_fwd_mask(xx, w3, 3, 1, 0.07, zc, 1)
        assert torch.equal(a, b), f"fwd mask, {c2} channels"


_CHUNK_CHILD = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from gansynth_amd import kernels
K = kernels.HipKernels()
d = torch.load(sys.argv[2])
CL = torch.channels_last
bad = []
for name, (x, wt, mask, ref_fwd, ref_t, ref_bd) in d.items():
    x, wt, mask = x.cuda().contiguous(memory_format=CL), wt.cuda(), mask.cuda().contiguous(memory_format=CL)
    if not torch.equal(K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1).cpu(), ref_fwd): bad.append(name + " fwd")
    if not torch.equal(K.conv2d_transpose_fwd(x, wt, 0.05).cpu(), ref_t): bad.append(name + " transposed")
    if not torch.equal(K.conv2d_bwd_data(x, wt, tuple(x.shape), 3, 1, 0.05, mask=mask, mask_act=1).cpu(), ref_bd): bad.append(name + " bwd_data+mask")
    if x.dtype == torch.bfloat16:   # the sign bits behind a leaky-relu result, written and read by chunked launches
        z = K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1)
        assert kernels._has_bits(z)
        zc = z.clone(memory_format=CL)
        assert not kernels._has_bits(zc)
        a = K.conv2d_bwd_data(x, wt, tuple(x.shape), 3, 1, 0.05, mask=z, mask_act=1)
        b = K.conv2d_bwd_data(x, wt, tuple(x.shape), 3, 1, 0.05, mask=zc, mask_act=1)
        if not torch.equal(a, b): bad.append(name + " bwd_data + 1-bit mask")
print("BAD", bad) if bad else print("CHUNKED-OK", len(d))
"""


def test_batches_beyond_the_item_decoders_range_run_in_image_chunks(K, tmp_path):
    """conv_igemm's division-free item decoder is exact below 2^21 work items; a larger launch (large-batch evaluation at full
    resolution: >= 4096 images of 128x1024 at 32 channels) is cut into launches of as many whole images as fit.  The limit is lowered
    (GS_IGEMM_MAX_ITEMS, read once per process: a child process) so that a small batch exercises it: forward (fused activation),
    transposed conv and a masked data gradient, bf16 and fp32, bit-identical to the single launch."""
    import subprocess
    import sys
    gen = torch.Generator(device="cuda").manual_seed(3)
    CL = torch.channels_last
    d = {}
    for name, dtype, (n, c, h, w) in (("bf16", torch.bfloat16, (7, 64, 8, 64)), ("f32", torch.float32, (5, 32, 8, 32)), ("bf16-32", torch.bfloat16, (6, 32, 16, 128))):
        x = torch.randn(n, c, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
        wt = torch.randn(3, 3, c, c, device="cuda", generator=gen)
        mask = torch.randn(n, c, h, w, device="cuda", generator=gen).to(dtype).contiguous(memory_format=CL)
        d[name] = (x.cpu(), wt.cpu(), mask.cpu(), K.conv2d_fwd_bias_act(x, wt, None, 3, 1, 0.05, 1).cpu(), K.conv2d_transpose_fwd(x, wt, 0.05).cpu(),
                   K.conv2d_bwd_data(x, wt, tuple(x.shape), 3, 1, 0.05, mask=mask, mask_act=1).cpu())
    path = str(tmp_path / "chunk_case.pt")
    torch.save(d, path)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = subprocess.run([sys.executable, "-c", _CHUNK_CHILD, root, path], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, GS_IGEMM_MAX_ITEMS="40"))
    assert res.returncode == 0 and "CHUNKED-OK 3" in res.stdout, (res.stdout[-800:], res.stderr[-1500:])
