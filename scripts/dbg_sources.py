import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables, kernels, _lib
dtype = torch.float32
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=False, dtype=dtype)
model.use_graphs = False
model.batch_d_tail = False
model.early_flush_always = True
K = kernels.get()
orig = K._flush_groups
names = {}
def flush(groups):
    for key, grp in groups.items():
        nm = names.get(grp["out"].data_ptr(), hex(grp["out"].data_ptr()))
        print("  flush: %-55s kind %s sources %d%s contiguous %s images %s" % (nm, key[0], len(grp["src"]), " > MAX" if len(grp["src"]) > _lib.WGRAD_MAX_SOURCES else "", grp["out"].is_contiguous(), [s[0].shape[0] for s in grp["src"]]))
    return orig(groups)
K._flush_groups = flush
lat, lab, real = R.synthetic_batch(4, rank=0, image_shape=(2, 16, 128))
lat, lab, real = cuda(lat), cuda(lab), cuda(real)
model._build(lat, lab)
for params in (model.d_params, model.g_params):
    for n, p in params.named.items():
        names[p.grad.data_ptr()] = n
print("D run"); model.discriminator_step(lat, lab, real)
print("G run"); model.generator_step(lat, lab)
