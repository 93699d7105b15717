"""Which torch-native device kernels (copies, adds, fills) does one training iteration launch, and from which line of gansynth_amd?"""
import collections, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict
from torch.profiler import ProfilerActivity, profile

dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
             discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
model = GANSynth(pggan.generator, pggan.discriminator, lambda: (pool[0][2], pool[0][1]), lambda: pool[0][0], None, hyper, dtype=dtype, use_graphs=False)
for _ in range(2):
    model.train_step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    model.train_step()
    torch.cuda.synchronize()
# CPU-side ops that launched a non-extension kernel, keyed by op name + innermost gansynth_amd frame
agg = collections.Counter()
dur = collections.Counter()
for ev in prof.events():
    if str(ev.device_type).endswith("CPU") and ev.kernels:
        for k in ev.kernels:
            if "gs::" in k.name or "gs_" in k.name:
                continue
            frame = next((f for f in (ev.stack or []) if "gansynth_amd" in f), "?")
            key = (ev.name, k.name[:40], str(ev.input_shapes)[:110] + " " + frame.strip()[-60:])
            agg[key] += 1
            dur[key] += k.duration
tot = 0
for key, n in sorted(agg.items(), key=lambda kv: -dur[kv[0]]):
    print("%3d x %7.1f us  %-16s %-42s %s" % (n, dur[key], key[0], key[1], key[2]))
    tot += dur[key]
print("total %.1f us in %d launches" % (tot, sum(agg.values())))
