"""Plain schedule vs the forked branches launched EAGERLY on real streams (reduced size, fp32): which gradient tensors differ after one
discriminator run / one generator run, and by how much?  (A difference beyond fp32 association is a missing stream dependency.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables

dtype = torch.float32
out = {}
for mode in ("plain", "forked-eager", "forked-eager-2"):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(1.0, variables.default_store(), full=False, dtype=dtype)
    model.use_graphs = False
    model.fork = mode != "plain"
    model.fork_eager = mode != "plain"
    model.early_flush_always = True
    model.batch_d_tail = False
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    res = []
    for step in range(int(os.environ.get("DBG_STEPS", "2"))):
        lat, lab, real = R.synthetic_batch(4, rank=step, image_shape=(2, 16, 128))
        lat, lab, real = cuda(lat).to(dtype), cuda(lab).to(dtype), cuda(real).to(dtype)
        if step == 0:
            model._build(lat, lab)
            variables.default_store().load_state_dict({**gp, **dp})
        model.discriminator_step(lat, lab, real)
        torch.cuda.synchronize()
        res.append({k: p.grad.clone() for k, p in model.d_params.named.items()})
        model.generator_step(lat, lab)
        torch.cuda.synchronize()
        res.append({k: p.grad.clone() for k, p in model.g_params.named.items()})
    out[mode] = res
for mode in ("forked-eager", "forked-eager-2"):
    for i, (a, b) in enumerate(zip(out["plain"], out[mode])):
        bad = []
        for k in a:
            d = float((a[k] - b[k]).abs().max()); s = float(a[k].abs().max()) + 1e-30
            if d > 1e-6 * s:
                bad.append((k, d / s))
        print(mode, "run", i, ("D" if i % 2 == 0 else "G"), "tensors beyond 1e-6:", len(bad), sorted(bad, key=lambda t: -t[1])[:6])
