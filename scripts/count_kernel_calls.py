"""Kernel-layer calls per training iteration by call site (CPU emulation of the kernel layer; small 3-level network, so counts
per level scale with the number of levels): which host line issues which kernel how often in the D run and in the G run."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict
from tests.cpu_kernels import CpuEmuKernels

counts = collections.Counter()
phase = ["?"]
flt = sys.argv[1] if len(sys.argv) > 1 else ""


class Wrapped(object):
    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, name):
        a = getattr(self._inner, name)
        if not callable(a):
            return a

        def call(*args, **kw):
            where = []
            for fr in reversed(traceback.extract_stack()[:-1]):
                if "/gansynth_amd/" in fr.filename and "kernels.py" not in fr.filename:
                    where.append(f"{os.path.basename(fr.filename)}:{fr.lineno}")
                    if len(where) == 2:
                        break
            counts[(phase[0], name, " <- ".join(where))] += 1
            return a(*args, **kw)
        return call


kernels.set_backend(Wrapped(CpuEmuKernels()))
variables.set_default_store(variables.VariableStore(device="cpu"))
RES = [int(v) for v in __import__("os").environ.get("GS_COUNT_RES", "8,64").split(",")]
pg = PGGAN(min_resolution=[2, 16], max_resolution=RES, min_channels=8, max_channels=16, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4,
             discriminator_beta1=0.0, discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0,
             fake_gradient_penalty_weight=0.0)
g = torch.Generator().manual_seed(0)
lat = torch.randn(4, 16, generator=g)
lab = torch.nn.functional.one_hot(torch.randint(0, 5, (4,), generator=g), 5).float()
img = torch.randn(4, 2, *RES, generator=g).clamp(-1, 1)
model = GANSynth(pg.generator, pg.discriminator, None, None, None, hyper)
model.discriminator_step(lat, lab, img)
model.generator_step(lat, lab)
counts.clear()
phase[0] = "D"
model._forward_backward("d", lat, lab, img)
phase[0] = "G"
model._forward_backward("g", lat, lab)
for (ph, name, where), c in sorted(counts.items(), key=lambda kv: (kv[0][0], kv[0][1], -kv[1])):
    if flt in name:
        print(f"{ph}  {c:3d}  {name:28s} {where}")
