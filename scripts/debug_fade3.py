import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import torch.nn.functional as TF
from oracle import torch_ref as R
from gansynth_amd import variables, ops
from gansynth_amd.networks import PGGAN
from gansynth_amd.variables import variable_scope

def relerr(got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    return float((got - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
def cuda(t):
    return t.cuda().contiguous(memory_format=torch.channels_last) if t.dim() == 4 else t.cuda()

kw = dict(min_resolution=[2, 16], max_resolution=[16, 128], min_channels=32, max_channels=64)
level = 0.12
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg = PGGAN(growing_level=level, **kw), R.PGGAN(growing_level=level, **kw)
lat, lab, real = R.synthetic_batch(4, rank=0, image_shape=(2, 16, 128))
gp, dp = opg.init_params(seed=0, bias_std=0.1)
with torch.no_grad():
    pg.generator(cuda(lat), cuda(lab)); pg.discriminator(cuda(real), cuda(lab))
st = variables.default_store(); st.load_state_dict({**gp, **dp})
names = ["discriminator/color_block_8x64/conv/weight", "discriminator/color_block_8x64/conv/bias", "discriminator/color_block_4x32/conv/weight", "discriminator/conv_block_8x64/conv/weight"]
with torch.no_grad():
    fake = pg.generator(cuda(lat), cuda(lab)); ofake = opg.generator(gp, lat, lab)
print("fake err", relerr(fake, ofake), "max abs diff", float((fake.cpu() - ofake).abs().max()))
odp = {k: v.clone().requires_grad_(True) for k, v in dp.items()}
for nm, img_g, img_o in [("real", cuda(real), real), ("fake(gpu G)", fake, ofake), ("fake(same images)", cuda(ofake), ofake)]:
    _, lg = pg.discriminator(img_g, cuda(lab)); loss = TF.softplus((lg * cuda(lab)).sum(1)).mean()
    g = torch.autograd.grad(loss, [st.variables[n] for n in names])
    _, olg = opg.discriminator(odp, img_o, lab); oloss = TF.softplus((olg * lab).sum(1)).mean()
    og = torch.autograd.grad(oloss, [odp[n] for n in names])
    print(nm, float(loss), float(oloss), [f"{relerr(a, b):.1e}" for a, b in zip(g, og)])
    # kink proximity of the colour-block pre-activation
    x0 = R.downscale2d(img_o, (2, 2)); z = R.conv2d(x0, dp[names[0]], dp[names[1]], (1, 1), 2.0)
    print("    min |z_color| (oracle)", float(z.abs().min()), " count |z|<1e-6:", int((z.abs() < 1e-6).sum()))
