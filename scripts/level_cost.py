import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables
dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs, model.keep_gradients = True, False
lat, lab, real = [cuda(t).to(dtype) for t in R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))]
model._build(lat, lab)
t0 = time.perf_counter()
model.discriminator_step(lat, lab, real); model.generator_step(lat, lab)
torch.cuda.synchronize()
print("two captures %.2f s, of which creating the throw-away streams %.3f s" % (time.perf_counter() - t0, model.level_seconds))
