"""Step time in every growing regime of the full-size schedule (eager launches in the fade-in regimes, hipGraph replay once fully
grown): which part of a real training run the headline number does not cover."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
from gansynth_amd import variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

dtype = torch.bfloat16
level = [0.0]
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pg = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=lambda: level[0])
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4,
             discriminator_beta1=0.0, discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0,
             fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
cur = [0]
model = GANSynth(pg.generator, pg.discriminator, lambda: (pool[cur[0] % len(pool)][2], pool[cur[0] % len(pool)][1]),
                 lambda: (cur.__setitem__(0, cur[0] + 1), pool[(cur[0] - 1) % len(pool)][0])[1], None, hyper, dtype=dtype, use_graphs=True)
full = 127.0
for depth in (0.0, 0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 6.5, 7.0):
    level[0] = float((2.0 ** depth - 1.0) / full) if depth < 7.0 else 1.0
    for _ in range(3):
        model.train_step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        model.train_step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n * 1e3
    print(f"growing_depth {pg.growing_depth:4.2f}: {dt:7.2f} ms per iteration ({8 / dt * 1e3:7.0f} images/s)  graphs: {sorted(model._graphs) or ('merged pair' if model._merged is not None else [])}")
