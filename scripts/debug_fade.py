import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from oracle import torch_ref as R
from gansynth_amd import variables
from gansynth_amd.networks import PGGAN
from gansynth_amd.models import GANSynth
from gansynth_amd.utils import Dict
import torch.nn.functional as TF

def relerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    return float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
def cuda(t):
    return t.cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.cuda()

kw = dict(min_resolution=[2, 16], max_resolution=[16, 128], min_channels=32, max_channels=64)
for level in (0.12, 0.1, 0.14, 0.05):
  for r1 in (0.0, 5.0):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg = PGGAN(growing_level=level, **kw), R.PGGAN(growing_level=level, **kw)
    hyper = dict(R.DEFAULT_HYPER); hyper["real_gradient_penalty_weight"] = r1
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, Dict(hyper))
    lat, lab, real = R.synthetic_batch(4, rank=0, image_shape=(2, 16, 128))
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    model._build(cuda(lat), cuda(lab))
    variables.default_store().load_state_dict({**gp, **dp})
    tr = R.Trainer(opg, gp, dp, hyper)
    # R1 gradient itself
    ri = cuda(real).requires_grad_(True)
    _, lg = pg.discriminator(ri, cuda(lab))
    rl = (lg * cuda(lab)).sum(1)
    g, = torch.autograd.grad(rl.sum(), ri)
    ro = real.clone().requires_grad_(True)
    _, olg = opg.discriminator(dp, ro, lab)
    og, = torch.autograd.grad((olg * lab).sum(), ro)
    d_loss = model.discriminator_step(cuda(lat), cuda(lab), cuda(real))
    d_grads = {k: p.grad.clone() for k, p in model.d_params.named.items()}
    od_loss, od_grads = tr.d_step(lat, lab, real)
    bad = {k: relerr(d_grads[k], od_grads[k]) for k in od_grads if float(od_grads[k].abs().max()) > 0}
    print(f"level {level} gd {pg.growing_depth:.3f} r1 {r1}: dlogit/dx err {relerr(g, og):.2e} loss {float(d_loss):.6f}/{float(od_loss):.6f} worst:", [(k.replace('discriminator/',''), f"{v:.1e}") for k, v in sorted(bad.items(), key=lambda kv: -kv[1])[:4]])
