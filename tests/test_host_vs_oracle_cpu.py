"""CPU: the host side (autograd algebra, ops surface, PGGAN wiring, losses, TF-Adam) against the
oracle, with the kernel layer emulated by tests/cpu_kernels.py (no GPU involved)."""
import numpy as np
import pytest
import torch

from oracle import torch_ref as R


def _small_pggan(level):
    from gansynth_amd.networks import PGGAN
    kw = dict(min_resolution=[2, 16], max_resolution=[8, 64], min_channels=8, max_channels=16)
    return PGGAN(growing_level=level, **kw), R.PGGAN(growing_level=level, **kw)


def _inputs(batch=4, latent=16, nlab=5, res=(8, 64), seed=0):
    g = torch.Generator().manual_seed(seed)
    lat = torch.randn(batch, latent, generator=g)
    lab = torch.nn.functional.one_hot(torch.randint(0, nlab, (batch,), generator=g), nlab).float()
    img = torch.randn(batch, 2, *res, generator=g).clamp(-1, 1)
    return lat, lab, img


def _oracle_params(opg, latent, nlab, seed=0):
    gen = torch.Generator().manual_seed(seed)
    out = []
    for shapes in opg.variable_shapes(latent_dim=latent, num_labels=nlab):
        d = {}
        for k, s in shapes.items():
            d[k] = torch.randn(s, generator=gen) * (0.1 if k.endswith("bias") else 1.0)
        out.append(d)
    return out


LEVELS = [0.0, 0.05, 1.0 / 7.0 + 1e-3, 0.3, 3.0 / 7.0, 0.6, 1.0]


@pytest.mark.parametrize("level", LEVELS)
def test_forward_matches_oracle(cpu_backend, level):
    from gansynth_amd import variables
    pg, opg = _small_pggan(level)
    lat, lab, img = _inputs()
    gp, dp = _oracle_params(opg, 16, 5)
    fake = pg.generator(lat, lab)
    feats, logits = pg.discriminator(img, lab)
    store = variables.default_store()
    assert list(store.trainable_variables("generator")) == list(gp)
    assert list(store.trainable_variables("discriminator")) == list(dp)
    store.load_state_dict({**gp, **dp})
    fake = pg.generator(lat, lab)
    feats, logits = pg.discriminator(img, lab)
    ofake = opg.generator(gp, lat, lab)
    ofeats, ologits = opg.discriminator(dp, img, lab)
    assert fake.shape == ofake.shape == (4, 2, 8, 64)
    torch.testing.assert_close(fake, ofake, rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(feats, ofeats, rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(logits, ologits, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("level,fake_penalty", [(0.0, 0.0), (0.3, 0.0), (1.0, 0.0), (0.3, 2.5)])
def test_losses_grads_and_adam_match_oracle(cpu_backend, level, fake_penalty):
    """`fake_penalty`: the optional zero-centred penalty on the generator distribution (models.py:50-54; weight 0 in
    gan_synth_main.py:87) -- tf.gradients(fake_logits, [fake_images]) differentiated once more into the discriminator's variables."""
    from gansynth_amd import variables
    from gansynth_amd.models import GANSynth
    from gansynth_amd.utils import Dict
    pg, opg = _small_pggan(level)
    lat, lab, img = _inputs()
    lat2, lab2, _ = _inputs(seed=1)
    gp, dp = _oracle_params(opg, 16, 5)
    hyper = Dict(R.DEFAULT_HYPER, fake_gradient_penalty_weight=fake_penalty)
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, hyper, keep_gradients=True)
    model._build(lat, lab)
    variables.default_store().load_state_dict({**gp, **dp})
    tr = R.Trainer(opg, gp, dp, hyper)
    if fake_penalty:   # the term is really there: the loss moves when it is switched off
        off = float(R.discriminator_loss(opg, gp, dp, lat, lab, img, dict(hyper, fake_gradient_penalty_weight=0.0)).detach())
        on = float(R.discriminator_loss(opg, gp, dp, lat, lab, img, dict(hyper)).detach())
        assert on > off + 1e-4 * abs(off), (on, off)

    d_loss = model.discriminator_step(lat, lab, img)
    d_grads = {k: p.grad.clone() for k, p in model.d_params.named.items()}
    od_loss, od_grads = tr.d_step(lat, lab, img)
    torch.testing.assert_close(d_loss, od_loss, rtol=1e-4, atol=1e-5)
    for k in od_grads:
        torch.testing.assert_close(d_grads[k], od_grads[k], rtol=2e-3, atol=2e-5, msg=lambda m, k=k: f"{k}: {m}")
    for k, p in model.d_params.named.items():
        torch.testing.assert_close(p.data, tr.d[k].data, rtol=1e-4, atol=1e-5, msg=lambda m, k=k: f"{k}: {m}")

    g_loss = model.generator_step(lat2, lab2)
    g_grads = {k: p.grad.clone() for k, p in model.g_params.named.items()}
    og_loss, og_grads = tr.g_step(lat2, lab2)
    torch.testing.assert_close(g_loss, og_loss, rtol=1e-4, atol=1e-5)
    for k in og_grads:
        scale = float(og_grads[k].abs().max()) + 1e-12
        torch.testing.assert_close(g_grads[k] / scale, og_grads[k] / scale, rtol=2e-3, atol=2e-4, msg=lambda m, k=k: f"{k}: {m}")
    for k, p in model.g_params.named.items():
        torch.testing.assert_close(p.data, tr.g[k].data, rtol=1e-4, atol=1e-5, msg=lambda m, k=k: f"{k}: {m}")
    assert model.global_step == tr.global_step == 1
