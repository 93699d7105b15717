"""(GPU) bf16 vs fp32 HIP step at the headline shape: per-tensor relative L2 / cosine / norm ratio of the gradients,
with the mode-seeking and R1 terms switched on and off (where does the bf16 error come from)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from oracle import torch_ref as R
from gansynth_amd import variables, kernels
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

B = int(os.environ.get("B", "8"))
res = (128, 1024)
kw = dict(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256)


def cuda(t):
    return t.cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.cuda()


def run(dtype, hyper, d_flat=None):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg = PGGAN(growing_level=1.0, **kw), R.PGGAN(growing_level=1.0, **kw)
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, Dict(hyper), dtype=dtype, keep_gradients=True)
    lat, lab, real = R.synthetic_batch(B, rank=0)
    lat2, lab2, _ = R.synthetic_batch(B, rank=1)
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    c = lambda t: cuda(t).to(dtype)
    model._build(c(lat), c(lab))
    variables.default_store().load_state_dict({**gp, **dp})
    d_loss = float(model.discriminator_step(c(lat), c(lab), c(real)))
    dg = {k: p.grad.clone() for k, p in model.d_params.named.items()}
    if d_flat is not None:
        with torch.no_grad():
            model.d_params.flat.copy_(d_flat)
        kernels.get().invalidate_weights(); kernels.get().refresh_weights()
    flat = model.d_params.flat.clone()
    g_loss = float(model.generator_step(c(lat2), c(lab2)))
    gg = {k: p.grad.clone() for k, p in model.g_params.named.items()}
    return d_loss, g_loss, dg, gg, flat


def report(tag, a, b):
    rows = []
    for k in b:
        x, y = a[k].double().flatten(), b[k].double().flatten()
        if float(y.norm()) == 0:
            continue
        rows.append((float((x - y).norm() / y.norm()), float(x @ y / (x.norm() * y.norm() + 1e-300)), float(x.norm() / y.norm()), k))
    rows.sort(reverse=True)
    print(f"--- {tag}: worst 8 of {len(rows)} (rel L2, cosine, norm ratio)")
    for r in rows[:8]:
        print("   %.4f  %.5f  %.4f  %s" % r)
    print("   median rel L2 %.4f" % sorted(r[0] for r in rows)[len(rows) // 2])


for name, ms, r1 in (("full losses", 0.1, 5.0), ("no mode-seeking", 0.0, 5.0), ("no R1", 0.1, 0.0)):
    hyper = dict(R.DEFAULT_HYPER, mode_seeking_loss_weight=ms, real_gradient_penalty_weight=r1)
    f = run(torch.float32, hyper)
    b = run(torch.bfloat16, hyper, d_flat=f[4])
    print(f"=== {name}: d_loss f32 {f[0]:.6f} bf16 {b[0]:.6f} | g_loss f32 {f[1]:.6f} bf16 {b[1]:.6f}")
    report("D grads", b[2], f[2])
    report("G grads", b[3], f[3])
