"""GANSynth driver on the MI355X path -- the command line of the reference's gan_synth_main.py (:26-37, :102-139).

    python gan_synth_main.py --train --model_dir gan_synth_model --filenames 'nsynth*.tfrecord' --batch_size 8 \
        --total_steps 1000000 --growing_steps 1000000
    python gan_synth_main.py --generate --model_dir gan_synth_model --filenames 'nsynth_test*.tfrecord'

Same flags and defaults; `--filenames` takes the reference's tfrecord files (or NSynth `examples.json` indexes) and
`--synthetic` replaces them by generated notes of the same shapes when no dataset is at hand.  Multi-GPU: launch with
`python -m torch.distributed.run --nproc-per-node N gan_synth_main.py ...` (one process per GPU, gradients all-reduced over
RCCL; the learning rates scale with the global batch like :79,82).  `--evaluate` needs the reference's frozen pitch-classifier
graph (a TensorFlow GraphDef, :111-122) and is not part of this path.
"""
import argparse
import glob
import os

import torch

parser = argparse.ArgumentParser()
parser.add_argument("--model_dir", type=str, default="gan_synth_model")
parser.add_argument("--filenames", type=str, default="nsynth*.tfrecord")
parser.add_argument("--batch_size", type=int, default=8)
parser.add_argument("--num_epochs", type=int, default=None)
parser.add_argument("--total_steps", type=int, default=1000000)
parser.add_argument("--growing_steps", type=int, default=1000000)
parser.add_argument("--classifier", type=str, default="pitch_classifier.pb")
parser.add_argument("--train", action="store_true")
parser.add_argument("--evaluate", action="store_true")
parser.add_argument("--generate", action="store_true")
# not in the reference
parser.add_argument("--synthetic", action="store_true", help="generated notes instead of --filenames")
parser.add_argument("--dtype", choices=["f32", "bf16"], default="bf16", help="activation storage (master weights / Adam stay fp32)")
parser.add_argument("--save_checkpoint_steps", type=int, default=1000)
parser.add_argument("--log_tensor_steps", type=int, default=100)
parser.add_argument("--num_generate_batches", type=int, default=None, help="stop --generate after this many batches (synthetic input never ends)")


def main(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    from gansynth_amd import checkpoint, variables
    from gansynth_amd.dataset import nsynth_input_fn, synthetic_nsynth_input_fn
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict

    torch.manual_seed(rank)   # tf.set_random_seed(0) (:44); one latent stream per rank
    dtype = torch.float32 if args.dtype == "f32" else torch.bfloat16
    device = torch.device("cuda", local_rank)
    variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
    pitches = range(24, 85)
    global_batch = args.batch_size * world

    holder = {}
    pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256,
                  growing_level=lambda: holder["model"].global_step / args.growing_steps)   # :52-55

    def real_input_fn_factory(train):
        if args.synthetic:
            return synthetic_nsynth_input_fn(args.batch_size, pitches, device=device, seed=rank,
                                             num_batches=None if train else args.num_generate_batches)
        files = sorted(glob.glob(args.filenames))
        if world > 1:
            files = files[rank::world] or files
        return nsynth_input_fn(files, args.batch_size, args.num_epochs if train else 1, shuffle=train, pitches=pitches,
                               sources=[0], device=device, seed=rank)

    real_input_fn = real_input_fn_factory(args.train)
    model = GANSynth(
        generator=pggan.generator, discriminator=pggan.discriminator,
        real_input_fn=real_input_fn,
        fake_input_fn=lambda: torch.randn(args.batch_size, 256, device=device),   # :70
        spectral_params=Dict(waveform_length=64000, sample_rate=16000, spectrogram_shape=[128, 1024], overlap=0.75),
        hyper_params=Dict(generator_learning_rate=8e-4 * global_batch / 8, generator_beta1=0.0, generator_beta2=0.99,
                          discriminator_learning_rate=8e-4 * global_batch / 8, discriminator_beta1=0.0, discriminator_beta2=0.99,
                          mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0),
        dtype=dtype, distributed=world > 1, use_graphs=True)
    holder["model"] = model

    if args.train:
        model.train(                                   # gan_synth_main.py:102-109, argument for argument
            model_dir=args.model_dir,                  # (every rank restores, rank 0 saves)
            config=None,                               # (the reference's tf.ConfigProto: nothing of it applies here)
            total_steps=args.total_steps,
            save_checkpoint_steps=args.save_checkpoint_steps,
            save_summary_steps=100,
            log_tensor_steps=args.log_tensor_steps,
            log=print if rank == 0 else None)
        if rank == 0:
            print(f"stopped at global_step = {model.global_step}")

    if args.evaluate:
        raise SystemExit("--evaluate needs the reference's TensorFlow pitch-classifier graph (gan_synth_main.py:111-122): not part of this path")

    if args.generate and rank == 0:
        from scipy.io import wavfile
        os.makedirs("samples", exist_ok=True)
        if not args.train:
            real_input_fn = model.real_input_fn
        num_waveforms, batches = 0, 0
        while args.num_generate_batches is None or batches < args.num_generate_batches:
            try:
                _, labels = real_input_fn()   # models.py:232-250: labels of the dataset, fresh latents
            except StopIteration:
                break
            latents = model.fake_input_fn()
            model._ensure_built(latents.to(dtype), labels.to(dtype))
            if batches == 0 and not args.train:
                path = checkpoint.restore(model, args.model_dir)
                print(f"restored {path}" if path else "no checkpoint found: generating from the initial weights")
            for waveform in model.generate(latents, labels).float().cpu().numpy():
                wavfile.write(f"samples/{num_waveforms}.wav", rate=16000, data=waveform)
                num_waveforms += 1
            batches += 1
        print(f"{num_waveforms} waveforms are generated in `samples` directory")

    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(parser.parse_args())
