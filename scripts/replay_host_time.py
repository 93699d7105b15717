"""Host time of the two graph replays of an iteration (no synchronisation in between) against the iteration's wall time: is the step launch-bound?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables

dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs, model.keep_gradients = True, False
lat, lab, real = [cuda(t).to(dtype) for t in R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))]
model._build(lat, lab)
model.real_input_fn, model.fake_input_fn = (lambda: (real, lab)), (lambda: lat)
step = (lambda: model.train_step()) if os.environ.get("RH_TRAIN_STEP", "1") != "0" else (lambda: (model.discriminator_step(lat, lab, real), model.generator_step(lat, lab)))
for _ in range(3):
    step()
torch.cuda.synchronize()
orig = torch.cuda.CUDAGraph.replay
host = []
def timed(self):
    t0 = time.perf_counter(); orig(self); host.append(time.perf_counter() - t0)
torch.cuda.CUDAGraph.replay = timed
N = 30
t0 = time.perf_counter()
for _ in range(N):
    step()
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print("fork" if model.fork else "plain", "merged" if model._merged is not None else "two runs", ": wall %.3f ms per iteration, host loop %.3f ms per iteration, graph.replay() host time %.3f ms mean (%d calls), max %.3f" %
      (t_all / N * 1e3, t_host / N * 1e3, sum(host) / len(host) * 1e3, len(host), max(host) * 1e3))
print("device memory: %.1f GB allocated at peak, %.1f GB reserved" % (torch.cuda.max_memory_allocated() / 1e9, torch.cuda.memory_reserved() / 1e9))
