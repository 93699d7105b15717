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
for level in (0.12, 0.1):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg = PGGAN(growing_level=level, **kw), R.PGGAN(growing_level=level, **kw)
    lat, lab, real = R.synthetic_batch(4, rank=0, image_shape=(2, 16, 128))
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    with variable_scope("discriminator"):
        pg._d_variables(61)
    variables.default_store().load_state_dict(dp)
    for p in variables.default_store().variables.values(): p.requires_grad_(True)
    head, fade = pg._head_depth(pg.growing_depth)
    img = cuda(real)
    with variable_scope("discriminator"):
        x0 = ops.downscale2d(img, (2, 2)); c2 = pg._d_color_block(x0, 2); b2 = pg._d_conv_block(c2, 2, 61)
        low = pg._d_color_block(ops.downscale2d(img, (4, 4)), 1); l = ops.lerp(low, b2, fade)
        b1 = pg._d_conv_block(l, 1, 61); feats, logits = pg._d_conv_block(b1, 0, 61)
    loss = TF.softplus(-(logits * cuda(lab)).sum(1)).mean()
    mine = torch.autograd.grad(loss, [c2, b2, low, l, b1])
    P = {k: v.clone().requires_grad_(True) for k, v in dp.items()}; n = "discriminator"
    def conv(x, s, st=(1, 1), vs=2.0): return R.conv2d(x, P[s + "/weight"], P[s + "/bias"], st, vs)
    ox0 = R.downscale2d(real, (2, 2)); oc2 = R.leaky_relu(conv(ox0, n + "/color_block_8x64/conv")); 
    t1 = R.leaky_relu(conv(oc2, n + "/conv_block_8x64/conv")); ob2 = R.leaky_relu(conv(t1, n + "/conv_block_8x64/conv_downscale", (2, 2)))
    olow = R.leaky_relu(conv(R.downscale2d(real, (4, 4)), n + "/color_block_4x32/conv")); ol = R.lerp(olow, ob2, fade)
    t2 = R.leaky_relu(conv(ol, n + "/conv_block_4x32/conv")); ob1 = R.leaky_relu(conv(t2, n + "/conv_block_4x32/conv_downscale", (2, 2)))
    for t in (oc2, ob2, olow, ol, ob1): t.retain_grad()
    # rest via the oracle's own discriminator tail is awkward; recompute block 0 inline
    s = n + "/conv_block_2x16"
    y = torch.cat([ob1, R.batch_stddev(ob1)], 1); y = R.leaky_relu(conv(y, s + "/conv")); y = y.reshape(4, -1)
    f = R.leaky_relu(R.dense(y, P[s + "/dense/weight"], P[s + "/dense/bias"], 2.0)); lg = R.dense(f, P[s + "/logits/weight"], P[s + "/logits/bias"], 1.0)
    oloss = TF.softplus(-(lg * lab).sum(1)).mean(); oloss.backward()
    print("level", level, "fade", fade, "loss", float(loss), float(oloss))
    for nm, a, b, fa, fb in zip(["c2", "b2", "low", "l", "b1"], mine, [oc2.grad, ob2.grad, olow.grad, ol.grad, ob1.grad], [c2, b2, low, l, b1], [oc2, ob2, olow, ol, ob1]):
        d = (a.float().cpu() - b).abs(); i = int(d.argmax())
        print(f"   {nm}: fwd err {relerr(fa, fb):.2e} grad err {relerr(a, b):.2e} argmax idx {np.unravel_index(i, d.shape)} got {float(a.float().cpu().flatten()[i]):.4e} ref {float(b.flatten()[i]):.4e} fwdval {float(fb.flatten()[i]):.4e}")
