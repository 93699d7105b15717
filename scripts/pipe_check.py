"""Step-by-step comparison of the sequential eager iteration, the pipelined one and the pipelined one with the update on a side
stream: parameters after every iteration must be bit-identical."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda
from oracle import torch_ref as R
from gansynth_amd import variables


runs = {}
for mode in ("eager", "pipe", "side", "side2"):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(1.0, variables.default_store(), full=False)
    model.use_graphs = model.pipeline = mode != "eager"
    model.pipe_side = mode.startswith("side")
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    batches = [R.synthetic_batch(4, rank=i, image_shape=(2, 16, 128)) for i in range(8)]
    cur = [0]

    def real_input_fn():
        lat, lab, real = batches[cur[0] % len(batches)]
        return cuda(real), cuda(lab)

    def fake_input_fn():
        lat, _, _ = batches[cur[0] % len(batches)]
        cur[0] += 1
        return cuda(lat)

    model.real_input_fn, model.fake_input_fn = real_input_fn, fake_input_fn
    lat, lab, _ = batches[0]
    model._build(cuda(lat), cuda(lab))
    variables.default_store().load_state_dict({**gp, **dp})
    trace = []
    for step in range(5):
        d_loss, g_loss = model.train_step()
        model.synchronize()
        torch.cuda.synchronize()
        trace.append((float(d_loss), float(g_loss), model.d_params.flat.clone(), model.g_params.flat.clone(), model.d_params.grad.clone(), model.g_params.grad.clone()))
    runs[mode] = trace
for mode in ("pipe", "side", "side2"):
    for i, (a, b) in enumerate(zip(runs["eager"], runs[mode])):
        print(mode, "step", i, "losses equal", a[0] == b[0], a[1] == b[1], "| D params", torch.equal(a[2], b[2]), "G params", torch.equal(a[3], b[3]),
              "| D grads", torch.equal(a[4], b[4]), int((a[4] != b[4]).sum()), "G grads", torch.equal(a[5], b[5]), int((a[5] != b[5]).sum()))
