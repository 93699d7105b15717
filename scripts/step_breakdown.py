"""Where a training iteration's wall time goes outside the two captured graphs: times (a) the full train_step loop,
(b) the two graph replays alone, (c) the eager tail (all-reduce-free Adam + weight-operand refresh) alone."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4,
             discriminator_beta1=0.0, discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0,
             fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
cur = [0]


def real_input_fn():
    lat, lab, real = pool[cur[0] % len(pool)]
    return real, lab


def fake_input_fn():
    lat, _, _ = pool[cur[0] % len(pool)]
    cur[0] += 1
    return lat


model = GANSynth(pggan.generator, pggan.discriminator, real_input_fn, fake_input_fn, None, hyper, dtype=dtype, use_graphs=True)
for _ in range(3):
    model.train_step()
torch.cuda.synchronize()


def timed(fn, n=20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


gd, gg = model._graphs["d"][0], model._graphs["g"][0]
hp = model.hyper_params
print(f"train_step              {timed(model.train_step):7.3f} ms")
print(f"D graph + G graph       {timed(lambda: (gd.replay(), gg.replay())):7.3f} ms")
print(f"D graph                 {timed(gd.replay):7.3f} ms")
print(f"G graph                 {timed(gg.replay):7.3f} ms")
print(f"Adam + refresh (D, G)   {timed(lambda: (model._apply(model.d_params, 8e-4, 0.0, 0.99), model._apply(model.g_params, 8e-4, 0.0, 0.99))):7.3f} ms")
t0 = time.perf_counter()
for _ in range(20):
    model.train_step()
cpu = (time.perf_counter() - t0) / 20 * 1e3
torch.cuda.synchronize()
print(f"CPU time to enqueue one train_step: {cpu:.3f} ms")
