"""CPU: the host-side rows of SURVEY.md 8(f) -- the NSynth input pipeline (reference dataset.py:12-91, read without
TensorFlow) and checkpoints keyed by the reference's variable names."""
import json
import os
import re
import struct
import wave

import numpy as np
import pytest
import torch

from gansynth_amd import dataset


def _varint(n):
    out = b""
    while True:
        b = n & 0x7F
        n >>= 7
        if n:
            out += bytes([b | 0x80])
        else:
            return out + bytes([b])


def _ld(field, payload):   # length-delimited protobuf field
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def _example(path, pitch, source, packed):
    """tf.train.Example{features{feature{"path": bytes_list, "pitch": int64_list, "source": int64_list}}}, written by hand."""
    def int_feature(v):
        v &= (1 << 64) - 1
        lst = _ld(1, _varint(v)) if packed else _varint((1 << 3) | 0) + _varint(v)
        return _ld(3, lst)
    feats = b""
    for name, feature in (("path", _ld(1, _ld(1, path.encode()))), ("pitch", int_feature(pitch)), ("source", int_feature(source))):
        feats += _ld(1, _ld(1, name.encode()) + _ld(2, feature))
    return _ld(1, feats)


def _write_tfrecord(filename, examples):
    with open(filename, "wb") as f:
        for ex in examples:
            f.write(struct.pack("<Q", len(ex)) + b"\0\0\0\0" + ex + b"\0\0\0\0")


def _write_wav(path, samples, channels=1):
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(2)
        w.setframerate(16000)
        w.writeframes(np.asarray(samples, dtype="<i2").tobytes())


@pytest.fixture
def nsynth_dir(tmp_path):
    rng = np.random.default_rng(0)
    recs = []
    for i in range(11):
        pitch = [20, 24, 30, 60, 84, 85, 40, 41, 42, 43, 44][i]
        source = [0, 0, 0, 1, 0, 0, 0, 0, 0, 0, 0][i]
        n = [64000, 64000, 1000, 64000, 70000, 64000, 64000, 64000, 64000, 64000, 64000][i]
        pcm = rng.integers(-32768, 32767, size=n)
        path = str(tmp_path / f"note{i}.wav")
        _write_wav(path, pcm)
        recs.append((path, pitch, source, pcm))
    _write_tfrecord(str(tmp_path / "nsynth_a.tfrecord"), [_example(p, pi, s, packed=False) for p, pi, s, _ in recs[:6]])
    _write_tfrecord(str(tmp_path / "nsynth_b.tfrecord"), [_example(p, pi, s, packed=True) for p, pi, s, _ in recs[6:]])
    return tmp_path, recs


def test_example_parser_reads_plain_and_packed_int64_lists():
    ex = dataset.parse_example(_example("/x/y.wav", 60, 2, packed=False))
    assert ex == {"path": b"/x/y.wav", "pitch": 60, "source": 2}
    ex = dataset.parse_example(_example("/x/y.wav", -3, 0, packed=True))
    assert ex["pitch"] == -3 and ex["source"] == 0


def test_nsynth_input_fn_filters_labels_batches_and_drops_the_remainder(nsynth_dir):
    tmp, recs = nsynth_dir
    files = sorted(str(p) for p in tmp.glob("*.tfrecord"))
    fn = dataset.nsynth_input_fn(files, batch_size=3, num_epochs=1, shuffle=False, pitches=range(24, 85), sources=[0])
    # kept (pitch in [24, 84], source 0), in file order: records 1, 2, 4, 6, 7, 8, 9, 10 -> two batches of 3, remainder (2) dropped
    kept = [1, 2, 4, 6, 7, 8, 9, 10]
    got = []
    while True:
        try:
            got.append(fn())
        except StopIteration:
            break
    assert len(got) == 2
    for b, (wav, lab) in enumerate(got):
        assert wav.shape == (3, 64000) and wav.dtype == torch.float32 and lab.shape == (3, 61)
        for j in range(3):
            path, pitch, source, pcm = recs[kept[3 * b + j]]
            ref = np.zeros(64000, dtype=np.float32)
            m = min(len(pcm), 64000)
            ref[:m] = pcm[:m].astype(np.float32) / 32768.0   # decode_wav: scaled to [-1, 1), cropped / zero-padded to 64000
            assert np.array_equal(wav[j].numpy(), ref)
            assert int(lab[j].argmax()) == pitch - 24 and float(lab[j].sum()) == 1.0
    with pytest.raises(StopIteration):
        fn()


def test_nsynth_input_fn_shuffles_and_repeats(nsynth_dir):
    tmp, _ = nsynth_dir
    files = sorted(str(p) for p in tmp.glob("*.tfrecord"))
    fn = dataset.nsynth_input_fn(files, batch_size=4, num_epochs=3, shuffle=True, pitches=range(24, 85), sources=[0], seed=1)
    labels = []
    while True:
        try:
            labels.append(fn()[1])
        except StopIteration:
            break
    assert len(labels) == (8 * 3) // 4    # 8 usable notes x 3 epochs, batches of 4 across epoch boundaries
    seen = torch.cat(labels).argmax(dim=1).tolist()
    assert sorted(seen) == sorted([p - 24 for p in (24, 30, 84, 40, 41, 42, 43, 44)] * 3)
    assert seen[:8] != seen[8:16] or seen[8:16] != seen[16:]   # reshuffled each epoch


def test_examples_json_index_and_stereo_wav(tmp_path):
    os.makedirs(tmp_path / "audio")
    _write_wav(str(tmp_path / "audio" / "guitar_001-060-100.wav"), np.stack([np.arange(100), -np.arange(100)], axis=1).reshape(-1), channels=2)
    with open(tmp_path / "examples.json", "w") as f:
        json.dump({"guitar_001-060-100": {"pitch": 60, "instrument_source": 0}}, f)
    fn = dataset.nsynth_input_fn(str(tmp_path / "examples.json"), batch_size=1, num_epochs=1, shuffle=False, pitches=range(24, 85), sources=[0])
    wav, lab = fn()
    assert np.array_equal(wav[0, :100].numpy(), np.arange(100, dtype=np.float32) / 32768.0) and float(wav[0, 100:].abs().max()) == 0.0
    assert int(lab[0].argmax()) == 36


def test_synthetic_notes_have_the_dataset_shapes_and_pitches():
    fn = dataset.synthetic_nsynth_input_fn(4, num_batches=2, seed=3)
    wav, lab = fn()
    assert wav.shape == (4, 64000) and lab.shape == (4, 61) and float(wav.abs().max()) <= 1.0
    k = int(lab[0].argmax())
    f0 = 440.0 * 2.0 ** ((24 + k - 69) / 12.0)
    spec = np.abs(np.fft.rfft(wav[0].numpy() * np.hanning(64000)))
    peak = np.fft.rfftfreq(64000, 1 / 16000.0)[spec.argmax()]
    assert min(abs(peak - f0 * h) for h in range(1, 5)) < 2.0   # the strongest partial is a harmonic of the labelled pitch
    fn()
    with pytest.raises(StopIteration):
        fn()


def test_checkpoint_round_trip_resumes_bit_identically(cpu_backend, tmp_path):
    """train(model_dir=...) saves variables, Adam slots, beta powers and global_step under the TF names; a fresh trainer that
    restores them continues exactly like the one that never stopped."""
    from safetensors.torch import load_file
    from gansynth_amd import checkpoint, variables
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict
    from oracle import torch_ref as R

    def make(seed):
        variables.set_default_store(variables.VariableStore(device="cpu", seed=seed))
        pg = PGGAN(min_resolution=[2, 16], max_resolution=[4, 32], min_channels=8, max_channels=16, growing_level=1.0)
        g = torch.Generator().manual_seed(5)
        batches = [(torch.randn(4, 16, generator=g), torch.nn.functional.one_hot(torch.randint(0, 5, (4,), generator=g), 5).float(),
                    torch.randn(4, 2, 4, 32, generator=g).clamp(-1, 1)) for _ in range(6)]
        cur = [0]

        def real_input_fn():
            return batches[cur[0] % 6][2], batches[cur[0] % 6][1]

        def fake_input_fn():
            cur[0] += 1
            return batches[(cur[0] - 1) % 6][0]

        return GANSynth(pg.generator, pg.discriminator, real_input_fn, fake_input_fn, None, Dict(R.DEFAULT_HYPER)), cur

    straight, _ = make(0)
    straight.train(total_steps=4, log=None)
    first, cur = make(0)
    first.train(total_steps=2, log=None, model_dir=str(tmp_path), save_checkpoint_steps=1)
    state = load_file(checkpoint.latest(str(tmp_path)))
    assert int(state["global_step"]) == 2 and os.path.basename(checkpoint.latest(str(tmp_path))) == "model.ckpt-2.safetensors"
    for name in ("generator/conv_block_4x32/upscale_conv/weight", "generator/conv_block_4x32/upscale_conv/weight/Adam",
                 "generator/conv_block_4x32/upscale_conv/weight/Adam_1", "discriminator/conv_block_2x16/logits/bias", "beta2_power", "beta2_power_1"):
        assert name in state, name
    assert tuple(state["generator/conv_block_4x32/upscale_conv/weight"].shape) == (3, 3, 16, 8)   # HWIO: [kh, kw, Cin=16, Cout=8]
    assert abs(float(state["beta2_power"]) - 0.99 ** 3) < 1e-7 and float(state["beta1_power_1"]) == 0.0
    second, cur2 = make(123)           # different initial weights: everything must come from the checkpoint
    cur2[0] = cur[0]                   # (the input position is not part of a TF checkpoint either: same data from here on)
    second.train(total_steps=4, log=None, model_dir=str(tmp_path), save_checkpoint_steps=0)
    assert second.restored_from.endswith("model.ckpt-2.safetensors") and second.global_step == 4
    assert torch.equal(second.g_params.flat, straight.g_params.flat) and torch.equal(second.d_params.flat, straight.d_params.flat)
    assert torch.equal(second.g_params.v, straight.g_params.v)
    assert checkpoint.latest(str(tmp_path)).endswith("model.ckpt-4.safetensors")
    # a checkpoint converted from a real tf.train.Saver file has no `optimizer_steps`: t is global_step (one apply per iteration)
    tf_like = {k: v for k, v in load_file(checkpoint.latest(str(tmp_path))).items() if not k.startswith("optimizer_steps")}
    third, _ = make(7)
    third._build(torch.zeros(4, 16), torch.zeros(4, 5))
    assert checkpoint.load_state_dict(third, tf_like, strict=True) == []
    assert (third.g_params.t, third.d_params.t, third.global_step) == (4, 4, 4)
    assert torch.equal(third.g_params.flat, straight.g_params.flat)
    # ... also where float32 beta2_power has underflowed (0.99^(t+1) is denormal past ~8.7 k steps, 0 past ~10.3 k): no restart of
    # the bias correction (lr_t would dip 10x); a beta2_power that contradicts global_step while still a normal float is an error
    for steps in (9000, 20000):
        late = dict(tf_like, global_step=torch.tensor(steps), beta2_power=torch.tensor(0.99 ** (steps + 1), dtype=torch.float32),
                    beta2_power_1=torch.tensor(0.99 ** (steps + 1), dtype=torch.float32))
        assert checkpoint.load_state_dict(third, late, strict=True) == []
        assert (third.g_params.t, third.d_params.t, third.global_step) == (steps, steps, steps)
    # while beta2_power is a normal float it IS the optimizer's step count: a checkpoint written between the two train ops of an
    # iteration (discriminator one step ahead, models.py:191-193) loads with t_D = global_step + 1 ...
    ahead = dict(tf_like, beta2_power_1=torch.tensor(0.99 ** 6, dtype=torch.float32))
    assert checkpoint.load_state_dict(third, ahead, strict=True) == []
    assert (third.g_params.t, third.d_params.t, third.global_step) == (4, 5, 4)
    # ... and one that contradicts global_step outright (a corrupt file, a beta2 that is not the checkpoint's: beta2_power says 4) is
    # refused; strict=False loads it loudly and keeps global_step -- the bias-correction step is never reset from a contradiction
    with pytest.raises(ValueError, match="says 4 optimizer steps"):
        checkpoint.load_state_dict(third, dict(tf_like, global_step=torch.tensor(400)), strict=True)
    with pytest.warns(UserWarning, match="says 4 optimizer steps"):
        checkpoint.load_state_dict(third, dict(tf_like, global_step=torch.tensor(400)), strict=False)
    assert (third.g_params.t, third.d_params.t, third.global_step) == (400, 400, 400)
    # TF accumulates beta2_power by float32 multiplies of float32(beta2): the recovered count stays within the accepted one step of
    # global_step far into a run (beta2 = 0.999: the double-precision log of the float32 product drifts past 1 after ~38 k steps)
    import numpy as np
    b2, p2 = np.float32(0.999), np.float32(1.0)
    steps = 30000
    for _ in range(steps + 1):
        p2 = np.float32(p2 * b2)
    old_b2 = (third.hyper_params.generator_beta2, third.hyper_params.discriminator_beta2)
    third.hyper_params.generator_beta2 = third.hyper_params.discriminator_beta2 = 0.999
    try:
        long_run = dict(tf_like, global_step=torch.tensor(steps), beta2_power=torch.tensor(float(p2)), beta2_power_1=torch.tensor(float(p2)))
        assert checkpoint.load_state_dict(third, long_run, strict=True) == []
        assert abs(third.g_params.t - steps) <= 1 and abs(third.d_params.t - steps) <= 1 and third.global_step == steps
    finally:
        third.hyper_params.generator_beta2, third.hyper_params.discriminator_beta2 = old_b2
    # a torn keep-forever clock file (a job killed mid-write) does not break the next save
    third.global_step = 4
    with open(os.path.join(str(tmp_path), "checkpoints_keep_clock"), "w") as f:
        f.write("")
    third.__dict__.pop("_saver_state", None)
    checkpoint.save(third, str(tmp_path))
    assert float(open(os.path.join(str(tmp_path), "checkpoints_keep_clock")).read()) > 0


def test_checkpoint_retention_follows_the_saver(cpu_backend, tmp_path):
    """tf.train.Saver(max_to_keep=10, keep_checkpoint_every_n_hours=12) (models.py:123-130): the newest `keep` files stay; a file
    leaving that window survives only if it was written after the saver's next keep-forever time, which then moves 12 h on."""
    from gansynth_amd import checkpoint, variables
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict
    from oracle import torch_ref as R
    variables.set_default_store(variables.VariableStore(device="cpu", seed=0))
    pg = PGGAN(min_resolution=[2, 16], max_resolution=[4, 32], min_channels=8, max_channels=16, growing_level=1.0)
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, Dict(R.DEFAULT_HYPER))
    model._build(torch.zeros(4, 16), torch.zeros(4, 5))
    hour = 3600.0
    for step, t in enumerate([0, 1, 2, 13, 14, 15, 16, 27], start=1):   # hours since the first save
        model.global_step = step
        checkpoint.save(model, str(tmp_path), keep=2, keep_every_n_hours=12.0, now=1000.0 + t * hour)
    left = sorted(int(re.findall(r"ckpt-(\d+)", f)[-1]) for f in os.listdir(tmp_path) if f.endswith(".safetensors"))
    # window = {7, 8}; step 4 (hour 13 > 12) was kept for good when it left the window, then the mark moved to hour 24:
    # steps 5, 6 (hours 14, 15) were deleted; step 8 (hour 27) is still inside the window
    assert left == [4, 7, 8], left
    assert open(os.path.join(tmp_path, "checkpoints_kept")).read().split() == ["model.ckpt-4.safetensors"]
    assert checkpoint.latest(str(tmp_path)).endswith("model.ckpt-8.safetensors")
    # a RESTARTED job (new trainer object, nothing remembered in memory): the files of the earlier session compete with their
    # modification times and the keep-forever clock (hour 24 by now) is read back -- step 8 (written at hour 27 by the first
    # session) is kept for good when it leaves the window, step 7 (hour 16) is deleted
    for f, t in ((7, 16), (8, 27)):
        os.utime(os.path.join(tmp_path, f"model.ckpt-{f}.safetensors"), (1000.0 + t * hour,) * 2)
    again = GANSynth(pg.generator, pg.discriminator, None, None, None, Dict(R.DEFAULT_HYPER))
    again.g_params, again.d_params = model.g_params, model.d_params
    for step, t in ((9, 28), (10, 29)):
        again.global_step = step
        checkpoint.save(again, str(tmp_path), keep=2, keep_every_n_hours=12.0, now=1000.0 + t * hour)
    left = sorted(int(re.findall(r"ckpt-(\d+)", f)[-1]) for f in os.listdir(tmp_path) if f.endswith(".safetensors"))
    assert left == [4, 8, 9, 10], left
    assert open(os.path.join(tmp_path, "checkpoints_kept")).read().split() == ["model.ckpt-4.safetensors", "model.ckpt-8.safetensors"]
