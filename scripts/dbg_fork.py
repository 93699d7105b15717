"""Which gradients of ONE iteration differ between the plain and the forked schedule (full size, bf16)?  usage: python scripts/dbg_fork.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables

dtype = torch.bfloat16
batches = [R.synthetic_batch(8, rank=i, image_shape=(2, 128, 1024)) for i in range(2)]
out = {}
for mode in ("plain", "forked", "forked2", "nomark", "nomark2", "eager", "eager2"):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
    model.use_graphs = not mode.startswith("eager")
    model.fork = mode != "plain"
    model.fork_eager = mode.startswith("eager")
    model.fork_marks = not mode.startswith("nomark")
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    lat, lab, real = batches[0]
    lat, lab, real = cuda(lat).to(dtype), cuda(lab).to(dtype), cuda(real).to(dtype)
    model._build(lat, lab)
    variables.default_store().load_state_dict({**gp, **dp})
    res = {}
    for rep in range(2):
        model.discriminator_step(lat, lab, real)
        res["d%d" % rep] = {k: p.grad.clone() for k, p in model.d_params.named.items()}
        model.generator_step(lat, lab)
        res["g%d" % rep] = {k: p.grad.clone() for k, p in model.g_params.named.items()}
    torch.cuda.synchronize()
    out[mode] = res
    del model
for a, b in (("forked", "forked2"), ("nomark", "nomark2"), ("forked", "nomark"), ("eager", "eager2"), ("forked", "eager")):
    for run in ("d0", "g0", "d1", "g1"):
        bad = []
        for k in out[a][run]:
            x, y = out[a][run][k], out[b][run][k]
            if not torch.equal(x, y):
                bad.append((k, float((x - y).abs().max()), float(x.abs().max())))
        print(a, "vs", b, run, ":", len(bad), "of", len(out[a][run]), "differ")
        for k, d, m in bad[-9:]:
            print("     %-70s max diff %.3e of %.3e" % (k, d, m))
