"""The launch chain at the turn-around of the discriminator's real pass (full size, bf16, eager, one stream): device kernels in start order with
duration and the gap to the previous one, from the last trunk conv of the forward to the first trunk data gradient of the R1 first-order pass.
usage: python scripts/dbg_tail_chain.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import ProfilerActivity, profile
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables

dtype = torch.bfloat16
lat, lab, real = R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs = False
model.fork_eager = False
model.real_input_fn = lambda: (cuda(real).to(dtype), cuda(lab).to(dtype))
model.fake_input_fn = lambda: cuda(lat).to(dtype)
gp, dp = opg.init_params(seed=0, bias_std=0.1)
model._build(cuda(lat).to(dtype), cuda(lab).to(dtype))
variables.default_store().load_state_dict({**gp, **dp})
for _ in range(2):
    model.train_step()
model.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    model.train_step()
    model.synchronize()
evs = sorted((e for e in prof.events() if str(e.device_type).endswith("CUDA")), key=lambda e: e.time_range.start)
t0 = evs[0].time_range.start
prev_end = t0
for i, e in enumerate(evs):
    s, d = e.time_range.start - t0, e.time_range.end - e.time_range.start
    print("%4d %9.1f us  dur %7.1f  gap %6.1f  %s" % (i, s, d, e.time_range.start - prev_end, e.name[:110]))
    prev_end = e.time_range.end
