import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1, device_id=torch.device("cuda", 0))
out = {}
for mode in ("plain", "dist+torch", "plain2", "dist"):
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(1.0, variables.default_store(), full=False)
    model.distributed, model.world, model.bucket_bytes = mode.startswith("dist"), 1, 16 << 10
    if mode == "dist+torch": os.environ["GS_TORCH_COLLECTIVES"] = "1"
    else: os.environ.pop("GS_TORCH_COLLECTIVES", None)
    model.use_graphs = False
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    rec = []
    for step in range(3):
        lat, lab, real = R.synthetic_batch(4, rank=step, image_shape=(2, 16, 128))
        if step == 0:
            model._build(cuda(lat), cuda(lab)); variables.default_store().load_state_dict({**gp, **dp})
        model.discriminator_step(cuda(lat), cuda(lab), cuda(real)); torch.cuda.synchronize()
        dg = model.d_params.grad.clone(); dpar = model.d_params.flat.clone()
        model.generator_step(cuda(lat), cuda(lab)); torch.cuda.synchronize()
        rec.append((dg, dpar, model.g_params.grad.clone(), model.g_params.flat.clone()))
    out[mode] = rec
for mode in ("dist+torch", "plain2", "dist"):
    for i in range(3):
        for k, nm in enumerate(("D grad", "D params", "G grad", "G params")):
            a, b = out["plain"][i][k], out[mode][i][k]
            d = (a - b).abs(); sc = float(a.abs().max())
            print("%-10s step %d %-8s max %.3g (scale %.3g) far %.4f" % (mode, i, nm, float(d.max()), sc, float((d > 1e-5 * sc).float().mean())))
dist.destroy_process_group()
