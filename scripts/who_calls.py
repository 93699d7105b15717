"""Which lines of gansynth_amd call a given kernel-layer method during one training iteration? usage: who_calls.py act_bwd [channel_sum ...]"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gansynth_amd import variables, kernels
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

names = sys.argv[1:] or ["act_bwd"]
dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
             discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
model = GANSynth(pggan.generator, pggan.discriminator, lambda: (pool[0][2], pool[0][1]), lambda: pool[0][0], None, hyper, dtype=dtype, use_graphs=False)
model.train_step()
K = kernels.get()
agg = collections.Counter()
for name in names:
    orig = getattr(K, name)

    def wrap(*a, _orig=orig, _name=name, **kw):
        st = [f for f in traceback.extract_stack()[:-1] if "gansynth_amd" in f.filename]
        shape = next((tuple(t.shape) for t in a if isinstance(t, torch.Tensor)), ())
        agg[(_name, " <- ".join("%s:%d(%s)" % (os.path.basename(f.filename), f.lineno, f.name) for f in st[-4:][::-1]), shape[1] if len(shape) > 1 else 0, torch.is_grad_enabled())] += 1
        return _orig(*a, **kw)

    setattr(K, name, wrap)
model.train_step()
torch.cuda.synchronize()
for key, n in sorted(agg.items(), key=lambda kv: -kv[1]):
    print("%3d x %-14s c=%-4d grad=%d  %s" % (n, key[0], key[2], key[3], key[1]))
