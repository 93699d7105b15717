"""Eager iterations of one growing regime (argv[1] = growing depth) for rocprofv3 --kernel-trace --stats."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from gansynth_amd import variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

depth = float(sys.argv[1])
dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pg = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=float((2.0 ** depth - 1.0) / 127.0))
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4,
             discriminator_beta1=0.0, discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0,
             fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
cur = [0]
model = GANSynth(pg.generator, pg.discriminator, lambda: (pool[cur[0] % len(pool)][2], pool[cur[0] % len(pool)][1]),
                 lambda: (cur.__setitem__(0, cur[0] + 1), pool[(cur[0] - 1) % len(pool)][0])[1], None, hyper, dtype=dtype, use_graphs=False)
for _ in range(4):
    model.train_step()
torch.cuda.synchronize()
