"""Two fresh trainers in one process must produce bit-identical losses and gradients (the second one runs on recycled
allocator blocks: a kernel that reads memory it never wrote shows up here).  GS_DEBUG_POISON_WS=1 turns such reads into NaN."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda
from oracle import torch_ref as R
from gansynth_amd import variables

outs = []
for rep in range(3):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(1.0, variables.default_store(), full=False)
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    losses, grads = [], []
    for step in range(2):
        lat, lab, real = R.synthetic_batch(4, rank=step, image_shape=(2, 16, 128))
        if step == 0:
            model._build(cuda(lat), cuda(lab))
            variables.default_store().load_state_dict({**gp, **dp})
        losses.append(float(model.discriminator_step(cuda(lat), cuda(lab), cuda(real))))
        grads.append({k: p.grad.clone() for k, p in model.d_params.named.items()})
        losses.append(float(model.generator_step(cuda(lat), cuda(lab))))
        grads.append({k: p.grad.clone() for k, p in model.g_params.named.items()})
    outs.append((losses, grads))
    del model
print("losses:", [o[0] for o in outs])
for rep in (1, 2):
    for i, (g0, g1) in enumerate(zip(outs[0][1], outs[rep][1])):
        bad = [(k, float((g0[k] - g1[k]).abs().max()), float(g0[k].abs().max())) for k in g0 if not torch.equal(g0[k], g1[k])]
        nan = [k for k in g1 if not torch.isfinite(g1[k]).all()]
        print(f"instance 0 vs {rep}, run {i} ({'DG'[i % 2]} step {i // 2}): {len(bad)} tensors differ, {len(nan)} non-finite", bad[:6], nan[:6])
