"""Which conv weight gradients of a training step are NOT deferred (and so miss the grouped launches)?  One stack per shape."""
import os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
             discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
model = GANSynth(pggan.generator, pggan.discriminator, lambda: (pool[0][2], pool[0][1]), lambda: pool[0][0], None, hyper, dtype=dtype, use_graphs=False)
K = kernels.get()
model.train_step()
seen = {}
for name in ("conv2d_bwd_weight", "conv2d_transpose_bwd_weight"):
    orig = getattr(K, name)
    def patched(x, gy, *a, _orig=orig, _name=name, **kw):
        if not (kw.get("out") is not None and K._pending is not None):
            key = (_name, tuple(x.shape), tuple(gy.shape), kw.get("out") is None)
            if key not in seen:
                seen[key] = 1
                print("NOT DEFERRED", key)
                print("".join(traceback.format_stack(limit=12)[:-1]))
        return _orig(x, gy, *a, **kw)
    setattr(K, name, patched)
model.train_step()
torch.cuda.synchronize()
print("done", len(seen))
