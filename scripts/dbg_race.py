"""Zero learning rates, so every replayed iteration must reproduce the eager gradients of its batch: which schedules ever do not?
usage: python scripts/dbg_race.py [iterations]   (DBG_KEEP=0: the zeroing optimizer step; DBG_FULL=1: full size bf16)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables
from gansynth_amd.utils import Dict
full = bool(os.environ.get("DBG_FULL"))
dtype = torch.bfloat16 if (full or os.environ.get("DBG_DTYPE") == "bf16") else torch.float32
n, res = (8, (2, 128, 1024)) if full else (4, (2, 16, 128))
NB = 3
batches = [R.synthetic_batch(n, rank=i, image_shape=res) for i in range(NB)]
hyper = Dict(R.DEFAULT_HYPER); hyper.generator_learning_rate = hyper.discriminator_learning_rate = 0.0
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
modes = os.environ.get("DBG_MODES", "eager,default,pair,sub_runs,nofork").split(",")
if any(m.startswith("dist") or m == "overlapped" for m in modes):
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1, device_id=torch.device("cuda", 0))
ref = None
for mode in modes:
    variables.set_default_store(variables.VariableStore(device="cuda"))
    pg, opg, model = make(float(os.environ.get("DBG_LEVEL", "1.0")), variables.default_store(), full=full, dtype=dtype, hyper=hyper)
    model.use_graphs, model.keep_gradients = mode not in ("eager", "sub_runs_eager"), not os.environ.get("DBG_NOKEEP")
    model.sub_runs = mode.startswith("sub_runs")
    if mode == "overlapped":
        mode_ = "dist_graph"; model.overlap_reduce = True
    if mode.startswith("dist") or mode == "overlapped":   # dist_eager / dist_torch_eager / dist_graph / overlapped
        model.distributed, model.world, model.bucket_bytes = True, 1, 16 << 10
        model.use_graphs = mode in ("dist_graph", "overlapped")
        if "torch" in mode: os.environ["GS_TORCH_COLLECTIVES"] = "1"
        else: os.environ.pop("GS_TORCH_COLLECTIVES", None)
    if mode == "sub_runs_nofork": model.fork = False
    if mode == "nofork": model.fork = False
    if mode == "pair": model.fuse_iteration = False
    cur = [0]
    def real_input_fn():
        lat, lab, real = batches[cur[0] % NB]; return cuda(real).to(dtype), cuda(lab).to(dtype)
    def fake_input_fn():
        lat, _, _ = batches[cur[0] % NB]; cur[0] += 1; return cuda(lat).to(dtype)
    model.real_input_fn, model.fake_input_fn = real_input_fn, fake_input_fn
    gp, dp = opg.init_params(seed=0, bias_std=0.1)
    lat, lab, _ = batches[0]
    model._build(cuda(lat).to(dtype), cuda(lab).to(dtype))
    variables.default_store().load_state_dict({**gp, **dp})
    rec, bad = [], 0
    period = None
    for it in range(iters if mode != "eager" else 2 * NB):
        model.train_step(); model.synchronize()
        g = (model.d_params.grad.clone(), model.g_params.grad.clone()) if model.keep_gradients else (model.d_params.v.clone(), model.g_params.v.clone())
        if mode == "eager":
            rec.append(g)
        else:
            r = ref[it % len(ref)]
            for k, nm in ((0, "D"), (1, "G")):
                e = float((g[k] - r[k]).abs().max()) / float(r[k].abs().max())
                if e > (1e-5 if dtype == torch.float32 else 1e-3):
                    bad += 1
                    names = [name for name, p in (model.d_params if k == 0 else model.g_params).named.items()
                             if float((p.grad - r[k][(p.grad.data_ptr() - (model.d_params if k == 0 else model.g_params).grad.data_ptr()) // 4:][:p.numel()].view(p.shape)).abs().max()) > 1e-4 * float(r[k].abs().max())]
                    print("  %s iter %d %s gradient off by %.3g: %s" % (mode, it, nm, e, names[:4]))
    if mode == "eager":
        ref = rec
        # the input functions advance one batch per iteration with period NB: iteration i of any mode sees what eager iteration i % period saw
        for i in range(NB):
            for k, nm in ((0, "D"), (1, "G")):
                d = float((ref[i][k] - ref[i + NB][k]).abs().max())
                if d != 0:
                    print("  eager iteration %d vs %d: %s gradient differs by %.3g of %.3g" % (i, i + NB, nm, d, float(ref[i][k].abs().max())))
        ref = ref[:NB]
    else:
        print("%s: %d of %d replayed gradient buffers differ from the eager ones" % (mode, bad, 2 * iters))
