"""What is each kernel family worth on the CRITICAL PATH of the replayed iteration?  The family's entry points are replaced by no-ops (outputs
stay uninitialised: the numbers are garbage, the schedule is the same minus those launches) and the iteration is timed again -- the difference
is the ceiling of anything that could be done to that family (fusing it away, making it free), in wall time rather than in summed kernel time.
usage (GPU box): python scripts/ablate_families.py [out file]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

FAMILIES = [
    ("nothing", ()),
    ("pixel norm, standalone launches", ("gs_pixel_norm",)),
    ("weight-gradient contraction (gs_conv_wgrad_jobs) + folds", ("gs_conv_wgrad_jobs", "gs_channel_fold_batch")),
    ("activation backward / bias sums / channel sums", ("gs_act_bwd", "gs_channel_sum", "gs_bias_act")),
    ("dense layers", ("gs_dense",)),
    ("1x1 colour convs", ("gs_conv2d_1x1", "gs_thin")),
    ("batch stddev + loss heads + sumsq / row scale / axpby", ("gs_batch_stddev", "gs_gan_", "gs_sumsq", "gs_row_scale", "gs_axpby")),
    ("Adam + operand refresh", ("gs_adam", "gs_weight_prep")),
    ("up / down scale, embedding", ("gs_upscale", "gs_blocksum", "gs_embedding")),
]


class SkipLib(object):
    def __init__(self, lib, prefixes):
        self.__dict__["_lib"], self.__dict__["_p"], self.__dict__["skipped"] = lib, tuple(prefixes), set()

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if self._p and name.startswith(self._p) and "workspace_bytes" not in name and "partial_rows" not in name:
            self.skipped.add(name)
            return lambda *a: 0
        return fn


dtype = torch.bfloat16
g = torch.Generator(device="cuda").manual_seed(1)
lab = torch.nn.functional.one_hot(torch.randint(0, 61, (8,), device="cuda", generator=g), 61).to(dtype)
img = torch.randn(8, 2, 128, 1024, device="cuda", generator=g).clamp(-1, 1).to(dtype).contiguous(memory_format=torch.channels_last)
hp = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
          discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
K = kernels.get()
real_lib = K.lib
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
base = None
for name, prefixes in FAMILIES:
    variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
    pg = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
    model = GANSynth(pg.generator, pg.discriminator, lambda: (img, lab), lambda: torch.randn(8, 256, device="cuda", dtype=dtype), None, hp, dtype=dtype, use_graphs=True)
    K.lib = SkipLib(real_lib, prefixes)
    try:
        for _ in range(6):
            model.train_step()
        model.synchronize()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(15):
                model.train_step()
            model.synchronize()
            best = min(best, (time.perf_counter() - t0) / 15 * 1e3)
    finally:
        skipped = sorted(K.lib.skipped)
        K.lib = real_lib
    if base is None:
        base = best
    print("%-62s %.3f ms  (%+.3f)   %s" % (name, best, best - base, ", ".join(s[3:] for s in skipped)[:150]), file=out, flush=True)
    del model
