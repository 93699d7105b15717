"""Checkpoints keyed by the reference's TensorFlow variable names.

A `tf.train.Saver` checkpoint of the reference graph (models.py:123-130) holds, by name:
  * every trainable variable under its scope name, e.g. `generator/conv_block_4x32/upscale_conv/weight`, conv filters in
    HWIO layout, dense weights [in, out] (ops.py:149-180) -- exactly how variables.VariableStore keeps them;
  * the Adam slots `<variable>/Adam` (m) and `<variable>/Adam_1` (v) (tf.train.AdamOptimizer slot names);
  * the non-slot accumulators of the two optimizers: `beta1_power`, `beta2_power` for the generator's (created first,
    models.py:81), `beta1_power_1`, `beta2_power_1` for the discriminator's (models.py:86);
  * `global_step` (int64).
The TF tensor-bundle container cannot be written without TensorFlow; the same names, shapes and layouts are stored in a
`.safetensors` file; `scripts/tf_checkpoint_convert.py` copies a TF checkpoint into that container (and back) on a machine
that has TensorFlow -- the only format gap.  `optimizer_steps[_1]` is an extra key (the exponent t itself);
a converted TF checkpoint lacks it and t is global_step (both optimizers step once per iteration; beta2_power = beta2^(t+1)
underflows in float32 past ~10 k steps and only cross-checks it).
Retention follows the reference's Saver: max_to_keep=10 and keep_checkpoint_every_n_hours=12 (models.py:126-127), see save().
"""
import glob
import math
import os
import re
import time
import warnings

import numpy as np
import torch

try:
    from safetensors.torch import load_file, save_file
except ImportError:   # pragma: no cover
    load_file = save_file = None


def state_dict(model):
    """name -> tensor (CPU) for everything a tf.train.Saver would write for `model` (a models.GANSynth after _build)."""
    if model.g_params is None:
        raise RuntimeError("checkpoint: the model has no variables yet (run a step or call _build first)")
    if hasattr(model, "synchronize"):
        model.synchronize()
    hp = model.hyper_params
    out = {}
    for params, suffix, b1, b2 in ((model.g_params, "", hp.generator_beta1, hp.generator_beta2),
                                   (model.d_params, "_1", hp.discriminator_beta1, hp.discriminator_beta2)):
        for name, p in params.named.items():
            n = p.numel()
            off = (p.data.data_ptr() - params.flat.data_ptr()) // 4
            out[name] = p.data.detach().cpu().clone()
            out[name + "/Adam"] = params.m[off:off + n].view(p.shape).detach().cpu().clone()
            out[name + "/Adam_1"] = params.v[off:off + n].view(p.shape).detach().cpu().clone()
        # tf.train.AdamOptimizer keeps beta^t as variables, multiplied once per apply_gradients (t = params.t here)
        out["beta1_power" + suffix] = torch.tensor(float(b1) ** (params.t + 1), dtype=torch.float32)
        out["beta2_power" + suffix] = torch.tensor(float(b2) ** (params.t + 1), dtype=torch.float32)
        out["optimizer_steps" + suffix] = torch.tensor(params.t, dtype=torch.int64)   # (not a TF variable: the exponent itself)
    out["global_step"] = torch.tensor(model.global_step, dtype=torch.int64)
    return out


def load_state_dict(model, state, strict=True):
    if model.g_params is None:
        raise RuntimeError("checkpoint: build the model's variables first (GANSynth._build)")
    if hasattr(model, "synchronize"):
        model.synchronize()
    # Two passes: everything that can refuse the file (shapes, missing entries, contradictory step counts) is checked BEFORE the first
    # byte is copied, so a refused checkpoint leaves the model exactly as it was (not half-restored with one optimizer's slots and the
    # other's step count).
    missing, copies, steps = [], [], {}
    for params, suffix in ((model.g_params, ""), (model.d_params, "_1")):
        for name, p in params.named.items():
            n = p.numel()
            off = (p.data.data_ptr() - params.flat.data_ptr()) // 4
            for key, dst in ((name, p.data), (name + "/Adam", params.m[off:off + n].view(p.shape)), (name + "/Adam_1", params.v[off:off + n].view(p.shape))):
                if key in state:
                    src = state[key]
                    if tuple(src.shape) != tuple(dst.shape):
                        raise ValueError(f"checkpoint: {key} has shape {tuple(src.shape)}, the variable has {tuple(dst.shape)}")
                    copies.append((dst, src))
                else:
                    missing.append(key)
        key = "optimizer_steps" + suffix
        if key in state:
            steps[suffix] = int(state[key])
        elif "global_step" in state:
            # a checkpoint converted from a real tf.train.Saver file has only TF's own accumulators.  Both optimizers are applied
            # exactly once per iteration (models.py:81-88, 191-192), so t IS global_step.  beta2_power = beta2^(t+1) (float32)
            # cannot give it back in general: 0.99^(t+1) is denormal past ~8.7 k steps and exactly 0 past ~10.3 k, where a
            # log() recovery would restart the bias correction (a 10x learning-rate dip); it only serves as a cross-check while
            # it is still a normal float.
            t = int(state["global_step"])
            hp = model.hyper_params
            b2 = float(hp.generator_beta2 if suffix == "" else hp.discriminator_beta2)
            p2 = float(state.get("beta2_power" + suffix, 0.0))
            if 1.2e-38 < p2 < 1.0 and 0.0 < b2 < 1.0:
                # While beta2_power is still a normal float it IS the optimizer's own step count (beta2^(t+1)): take t from it -- a
                # reference checkpoint written between the discriminator's and the generator's train op (OutOfRangeError after the
                # first session.run of models.py:191-192) has t_D = global_step + 1.  A disagreement within 0.1 % is float32 product
                # drift (TF accumulates beta2_power by float32 multiplies of float32(beta2): the log is taken of THAT base) and
                # global_step stands; a gross one -- a corrupt file, a beta2 hyper-parameter that is not the checkpoint's, or
                # optimizers that really ran different numbers of steps -- is REFUSED under strict=True (the default of restore()
                # and train()) and a warning under strict=False: the Adam bias-correction step is not silently reset from it.
                t_log = int(round(math.log(p2) / math.log(float(np.float32(b2))))) - 1
                if abs(t_log - t) <= 1:
                    t = max(t_log, 0)
                elif abs(t_log - t) > max(1, int(1e-3 * t)):
                    msg = (f"checkpoint: beta2_power{suffix} = {p2:g} says {t_log} optimizer steps with beta2 = {b2:g}, "
                           f"global_step says {t}")
                    if strict:
                        raise ValueError(msg + " (strict=False keeps global_step)")
                    warnings.warn(msg + "; keeping global_step")
            steps[suffix] = t
        else:
            missing.append(key)
    if "global_step" not in state:
        missing.append("global_step")
    if strict and missing:
        raise KeyError(f"checkpoint: {len(missing)} entries missing, e.g. {missing[:4]}")
    for dst, src in copies:
        dst.copy_(src.to(dst.device, dst.dtype))
    for params, suffix in ((model.g_params, ""), (model.d_params, "_1")):
        if suffix in steps:
            params.t = steps[suffix]
    if "global_step" in state:
        model.global_step = int(state["global_step"])
    from . import kernels
    K = kernels.get()
    if hasattr(K, "invalidate_weights"):   # the conv kernels' re-laid weight operands are stale now; captured graphs expect fresh ones
        K.invalidate_weights()
        K.refresh_weights()
    return missing


def save(model, model_dir, keep=10, keep_every_n_hours=12.0, now=None):
    """`model_dir/model.ckpt-<global_step>.safetensors` (+ a `checkpoint` text file naming the latest, like tf.train.Saver).
    Retention as tf.train.Saver(max_to_keep=10, keep_checkpoint_every_n_hours=12) (models.py:123-130): the newest `keep` files stay;
    a file that falls out of that window is deleted UNLESS it was written later than the saver's next keep-forever time, in which
    case it is kept for good (its name goes to `checkpoints_kept`) and that time moves `keep_every_n_hours` on.  Unlike a Saver
    object, which forgets everything at a restart, the keep-forever clock is persisted (`checkpoints_keep_clock`) and files of earlier
    sessions compete with their modification times, so that a job restarted often still keeps long-term checkpoints.  `now`: clock
    override for tests."""
    if save_file is None:
        raise RuntimeError("checkpoint: the safetensors package is not importable")
    now = time.time() if now is None else now
    os.makedirs(model_dir, exist_ok=True)
    path = os.path.join(model_dir, f"model.ckpt-{model.global_step}.safetensors")
    save_file({k: v.contiguous() for k, v in state_dict(model).items()}, path)
    with open(os.path.join(model_dir, "checkpoint"), "w") as f:
        f.write(f'model_checkpoint_path: "{os.path.basename(path)}"\n')
    kept_file, clock_file = os.path.join(model_dir, "checkpoints_kept"), os.path.join(model_dir, "checkpoints_keep_clock")
    kept = set(open(kept_file).read().split()) if os.path.exists(kept_file) else set()
    saver = model.__dict__.setdefault("_saver_state", {"next_keep": None, "times": {}})
    if saver["next_keep"] is None and keep_every_n_hours:
        # the keep-forever clock survives restarts (a job restarted more often than every keep_every_n_hours would otherwise never
        # keep a long-term checkpoint)
        saver["next_keep"] = now + keep_every_n_hours * 3600.0
        if os.path.exists(clock_file):
            try:
                saver["next_keep"] = float(open(clock_file).read())
            except (OSError, ValueError):   # an empty / torn clock file (a job killed mid-write): start the clock again
                pass
        with open(clock_file, "w") as f:
            f.write(repr(saver["next_keep"]))
    saver["times"][os.path.basename(path)] = now
    old = sorted((p for p in glob.glob(os.path.join(model_dir, "model.ckpt-*.safetensors")) if os.path.basename(p) not in kept),
                 key=lambda p: int(re.findall(r"ckpt-(\d+)", p)[-1]))
    for p in old[:-keep] if keep else []:
        name = os.path.basename(p)
        written = saver["times"].get(name)
        if written is None:   # written by an earlier session of this run: its modification time stands in
            written = os.path.getmtime(p)
        if saver["next_keep"] is not None and written > saver["next_keep"]:
            kept.add(name)
            saver["next_keep"] += keep_every_n_hours * 3600.0
            with open(kept_file, "w") as f:
                f.write("\n".join(sorted(kept)) + "\n")
            with open(clock_file, "w") as f:
                f.write(repr(saver["next_keep"]))
        else:
            os.remove(p)
    return path


def latest(model_dir):
    marker = os.path.join(model_dir, "checkpoint")
    if os.path.exists(marker):
        m = re.search(r'model_checkpoint_path: "([^"]+)"', open(marker).read())
        if m and os.path.exists(os.path.join(model_dir, m.group(1))):
            return os.path.join(model_dir, m.group(1))
    files = glob.glob(os.path.join(model_dir, "model.ckpt-*.safetensors"))
    return max(files, key=lambda p: int(re.findall(r"ckpt-(\d+)", p)[-1])) if files else None


def restore(model, model_dir_or_file, strict=True):
    """Load the latest checkpoint of a directory (or a given file); returns its path, or None when there is none
    (a fresh run, like tf.train.MonitoredSession with an empty checkpoint_dir)."""
    if load_file is None:
        raise RuntimeError("checkpoint: the safetensors package is not importable")
    path = model_dir_or_file if os.path.isfile(model_dir_or_file) else latest(model_dir_or_file)
    if path is None:
        return None
    load_state_dict(model, load_file(path), strict=strict)
    return path
