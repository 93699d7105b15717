"""The reference's own call sites run unchanged against this package (VERDICT r5 "missing" item 4): the keyword sets below are copied
from the CALL SITES -- /root/reference/gan_synth_main.py:102-109 (train), :128-131 (generate) -- and from the DEFINITIONS of the layer
functions, /root/reference/ops.py:149-154,183-189,204-209,221-229,250-258 (argument names and order).  CPU backend, reduced network."""
import inspect

import numpy as np
import pytest
import torch


def _model(tmp_path, seed=0, batches=3):
    from gansynth_amd import variables
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict
    from oracle import torch_ref as R

    variables.set_default_store(variables.VariableStore(device="cpu", seed=seed))
    pg = PGGAN(min_resolution=[2, 16], max_resolution=[4, 32], min_channels=8, max_channels=16, growing_level=1.0)
    g = torch.Generator().manual_seed(5)
    data = [(torch.randn(4, 16, generator=g), torch.nn.functional.one_hot(torch.randint(0, 5, (4,), generator=g), 5).float(),
             torch.randn(4, 2, 4, 32, generator=g).clamp(-1, 1)) for _ in range(batches)]
    cur = [0, 0]

    def real_input_fn():
        if cur[0] >= 2 * batches:
            raise StopIteration   # tf.errors.OutOfRangeError of a one-shot iterator (models.py:193,249)
        cur[0] += 1
        return data[(cur[0] - 1) % batches][2], data[(cur[0] - 1) % batches][1]

    def fake_input_fn():
        cur[1] += 1
        return data[(cur[1] - 1) % batches][0]

    return GANSynth(pg.generator, pg.discriminator, real_input_fn, fake_input_fn, None, Dict(R.DEFAULT_HYPER)), cur


def test_train_takes_the_reference_call(cpu_backend, tmp_path):
    """gan_synth_main.py:102-109, keyword for keyword (config: a stand-in object for the tf.ConfigProto of :91-98)."""
    from gansynth_amd import checkpoint
    model, _ = _model(tmp_path)
    config = object()
    model.train(
        model_dir=str(tmp_path),
        config=config,
        total_steps=2,
        save_checkpoint_steps=1000,
        save_summary_steps=100,
        log_tensor_steps=100,
    )
    assert model.global_step == 2
    assert checkpoint.latest(str(tmp_path)).endswith("model.ckpt-2.safetensors")
    # positional, in the order of the definition (models.py:110)
    again, _ = _model(tmp_path)
    again.train(str(tmp_path), config, 3, 1000, 100, 100)
    assert again.restored_from.endswith("model.ckpt-2.safetensors") and again.global_step == 3
    names = list(inspect.signature(model.train).parameters)[:6]
    assert names == ["model_dir", "config", "total_steps", "save_checkpoint_steps", "save_summary_steps", "log_tensor_steps"]
    with pytest.raises(TypeError):
        model.train(model_dir=str(tmp_path))   # total_steps has no default in the reference either


def test_train_still_takes_the_step_count_first(cpu_backend, tmp_path):
    model, _ = _model(tmp_path)
    model.train(2, log=None)
    assert model.global_step == 2


def test_generate_takes_the_reference_call(cpu_backend, tmp_path, monkeypatch):
    """gan_synth_main.py:128-136: `for waveforms in gan_synth.generate(model_dir=, config=): for waveform in waveforms: wavfile.write`
    -- a generator of numpy batches that restores the checkpoint of model_dir and ends when the input runs dry."""
    from gansynth_amd import spectral_ops
    trained, _ = _model(tmp_path)
    trained.train(model_dir=str(tmp_path), config=None, total_steps=1, save_checkpoint_steps=1, save_summary_steps=1, log_tensor_steps=1, log=None)
    # the reduced network's images are not 128 x 1024: the inverse transform is not what is under test here
    monkeypatch.setattr(spectral_ops, "convert_images_to_waveform", lambda images, **kw: images.reshape(images.shape[0], -1).float())
    fresh, cur = _model(tmp_path, seed=99)   # other initial weights: what comes out must be the checkpoint's generator
    fresh.spectral_params = {}
    out = fresh.generate(
        model_dir=str(tmp_path),
        config=None,
    )
    assert inspect.isgenerator(out)
    batches = list(out)
    assert len(batches) == 6 and all(isinstance(b, np.ndarray) and b.shape == (4, 2 * 4 * 32) and b.dtype == np.float32 for b in batches)
    assert fresh.restored_from.endswith("model.ckpt-1.safetensors")
    assert torch.equal(fresh.g_params.flat, trained.g_params.flat)
    # the tensor form of rounds 1-5 is still there
    lat, lab = torch.randn(4, 16), torch.nn.functional.one_hot(torch.tensor([0, 1, 2, 3]), 5).float()
    assert torch.is_tensor(fresh.generate(lat, lab))


def test_layer_functions_take_the_reference_arguments(cpu_backend):
    """ops.py:149-154,183-189,204-209,221-229,250-258: the reference's argument names in the reference's order; the two weight
    normalisers are accepted as False and refused -- not ignored -- as True."""
    from gansynth_amd import ops, variables
    ref_order = {
        "get_weight": ["shape", "variance_scale", "scale_weight", "apply_weight_standardization", "apply_spectral_normalization"],
        "dense": ["inputs", "units", "use_bias", "variance_scale", "scale_weight", "apply_weight_standardization", "apply_spectral_normalization"],
        "embedding": ["inputs", "units", "variance_scale", "scale_weight", "apply_weight_standardization", "apply_spectral_normalization"],
        "conv2d": ["inputs", "filters", "kernel_size", "strides", "use_bias", "variance_scale", "scale_weight", "apply_weight_standardization",
                   "apply_spectral_normalization"],
        "conv2d_transpose": ["inputs", "filters", "kernel_size", "strides", "use_bias", "variance_scale", "scale_weight",
                             "apply_weight_standardization", "apply_spectral_normalization"],
    }
    for name, order in ref_order.items():
        assert list(inspect.signature(getattr(ops, name)).parameters)[:len(order)] == order, name
    variables.set_default_store(variables.VariableStore(device="cpu", seed=0))
    x = torch.randn(2, 4, 4, 8).contiguous(memory_format=torch.channels_last)
    with variables.variable_scope("a"):
        y = ops.conv2d(
            inputs=x,
            filters=6,
            kernel_size=[3, 3],
            strides=[1, 1],
            use_bias=True,
            variance_scale=2.0,
            scale_weight=True,
            apply_weight_standardization=False,
            apply_spectral_normalization=False,
        )
    assert tuple(y.shape) == (2, 6, 4, 8)
    with variables.variable_scope("b"):
        z = ops.dense(torch.randn(2, 5), 3, True, 2.0, True, False, False)
        e = ops.embedding(torch.eye(5)[:2], 3, 1.0, True, False, False)
    assert tuple(z.shape) == (2, 3) and tuple(e.shape) == (2, 3)
    for flag in ("apply_weight_standardization", "apply_spectral_normalization"):
        for call in (lambda **k: ops.conv2d(x, 6, [3, 3], **k), lambda **k: ops.conv2d_transpose(x, 6, [3, 3], [2, 2], **k),
                     lambda **k: ops.dense(torch.randn(2, 5), 3, **k), lambda **k: ops.embedding(torch.eye(5)[:2], 3, **k),
                     lambda **k: ops.get_weight([5, 3], **k)):
            with variables.variable_scope("c"), pytest.raises(NotImplementedError):
                call(**{flag: True})
