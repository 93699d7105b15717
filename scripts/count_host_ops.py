"""Which torch-native ops (each one a small GPU launch on the product path) the host code issues per training iteration outside
the kernel layer.  Runs one D run + one G run on the CPU emulation of the kernel layer (tests/cpu_kernels.py) under a
TorchDispatchMode and groups the aten ops by the gansynth_amd source line that issued them.  CPU only; no GPU needed."""
import collections
import os
import sys
import traceback

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.utils._python_dispatch import TorchDispatchMode

from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict
from tests.cpu_kernels import CpuEmuKernels

inside = [0]


class Wrapped(object):
    def __init__(self, inner):
        self._inner = inner

    def __getattr__(self, name):
        a = getattr(self._inner, name)
        if not callable(a):
            return a

        def call(*args, **kw):
            inside[0] += 1
            try:
                return a(*args, **kw)
            finally:
                inside[0] -= 1
        return call


counts = collections.Counter()
SKIP = ("aten.view", "aten._unsafe_view", "aten.detach", "aten.alias", "aten.t.", "aten.transpose", "aten.permute", "aten.expand", "aten.squeeze",
        "aten.unsqueeze", "aten.select", "aten.slice", "aten.as_strided", "aten.empty", "aten.reshape", "aten.is_", "aten.stride", "aten.size",
        "aten.sym_", "aten._local_scalar", "aten.lift_fresh", "aten.narrow", "aten.unbind", "aten.split", "aten.new_empty", "aten.result_type")


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        out = func(*args, **(kwargs or {}))
        name = str(func)
        if inside[0] == 0 and not name.startswith(SKIP):
            where = "?"
            for fr in reversed(traceback.extract_stack()):
                if "/gansynth_amd/" in fr.filename:
                    where = f"{os.path.basename(fr.filename)}:{fr.lineno} {fr.line.strip()[:70]}"
                    break
            numel = 0
            for a in list(args) + [out]:
                if isinstance(a, torch.Tensor):
                    numel = max(numel, a.numel())
            counts[(where, name, "big" if numel > 4096 else "small")] += 1
        return out


kernels.set_backend(Wrapped(CpuEmuKernels()))
variables.set_default_store(variables.VariableStore(device="cpu"))
pg = PGGAN(min_resolution=[2, 16], max_resolution=[8, 64], min_channels=8, max_channels=16, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4,
             discriminator_beta1=0.0, discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0,
             fake_gradient_penalty_weight=0.0)
g = torch.Generator().manual_seed(0)
lat = torch.randn(4, 16, generator=g)
lab = torch.nn.functional.one_hot(torch.randint(0, 5, (4,), generator=g), 5).float()
img = torch.randn(4, 2, 8, 64, generator=g).clamp(-1, 1)
DT = torch.bfloat16 if __import__("os").environ.get("GS_COUNT_BF16") else torch.float32
model = GANSynth(pg.generator, pg.discriminator, None, None, None, hyper, dtype=DT)
lat, lab, img = lat.to(DT), lab.to(DT), img.to(DT)
model.discriminator_step(lat, lab, img)
model.generator_step(lat, lab)
with Log():
    model._forward_backward("d", lat, lab, img)
    model._forward_backward("g", lat, lab)
tot = sum(counts.values())
print(f"{tot} torch-native ops per iteration outside the kernel layer (fwd+bwd of both runs; autograd-engine ops show up at the line "
      f"whose backward triggered them)")
for (where, name, size), c in sorted(counts.items(), key=lambda kv: -kv[1]):
    print(f"{c:4d}  {size:5s} {name:40s} {where}")
