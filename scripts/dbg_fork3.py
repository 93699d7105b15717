"""In-replay experiment: every pixel_norm_bwd_bwd of the forked generator-run graph runs twice (on its operands, then on copies of them taken
before the first run) and its operands are copied again afterwards: do the operands change under the kernel, do two runs agree?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables, kernels

dtype = torch.bfloat16
lat, lab, real = R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs = True
gp, dp = opg.init_params(seed=0, bias_std=0.1)
lat, lab, real = cuda(lat).to(dtype), cuda(lab).to(dtype), cuda(real).to(dtype)
model._build(lat, lab)
variables.default_store().load_state_dict({**gp, **dp})
K = kernels.get()
rec = []
orig = type(K).pixel_norm_bwd_bwd
def twice(self, gg, g, x, eps, pre_act=0, with_g=False):
    if not torch.cuda.is_current_stream_capturing():
        return orig(self, gg, g, x, eps, pre_act=pre_act, with_g=with_g)
    before = [t.clone() for t in (gg, g, x)]
    out = orig(self, gg, g, x, eps, pre_act=pre_act, with_g=with_g)
    o1 = [t.clone() for t in (out if with_g else (out,))]
    after = [t.clone() for t in (gg, g, x)]
    out2 = orig(self, before[0], before[1], before[2], eps, pre_act=pre_act, with_g=with_g)
    o2 = [t.clone() for t in (out2 if with_g else (out2,))]
    out3 = orig(self, gg, g, x, eps, pre_act=pre_act, with_g=with_g)
    o3 = [t.clone() for t in (out3 if with_g else (out3,))]
    rec.append((tuple(x.shape), before, after, o1, o2, o3))
    return out
type(K).pixel_norm_bwd_bwd = twice
for rep in range(4):
    model._run("g", lat, lab)
    torch.cuda.synchronize()
    print("replay", rep)
    for i, (shape, before, after, o1, o2, o3) in enumerate(rec):
        ops = [bool(torch.equal(a, b)) for a, b in zip(before, after)]
        r12 = [bool(torch.equal(a, b)) for a, b in zip(o1, o2)]
        r13 = [bool(torch.equal(a, b)) for a, b in zip(o1, o3)]
        if not (all(ops) and all(r12) and all(r13)):
            nd = [int((a != b).sum()) for a, b in zip(o1, o3)]
            a, b = o1[0].permute(0, 2, 3, 1).reshape(-1, o1[0].shape[1]).float(), o3[0].permute(0, 2, 3, 1).reshape(-1, o1[0].shape[1]).float()
            rows = (a != b).any(dim=1).nonzero().flatten().tolist()
            per_row = [(r, int((a[r] != b[r]).sum()), float((a[r] - b[r]).abs().max() / (a[r].abs().max() + 1e-30))) for r in rows[:12]]
            print("      rows differing (1 vs 3): %d of %d; first (row, channels differing, rel diff): %s" % (len(rows), a.shape[0], per_row))
            print("   call %d %s: operands unchanged %s, run on operands == run on copies %s, == third run %s (elements differing %s)" % (i, shape, ops, r12, r13, nd))
