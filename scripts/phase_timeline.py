"""Phase boundaries of the replayed iteration, measured from INSIDE the graph (scripts/probe/stamp.hip: a one-thread kernel writes the
constant 100 MHz clock), i.e. without rocprofv3 / roctracer between the host and the queues.  ~25 extra nodes in a 406-node graph.
usage (GPU box): python scripts/phase_timeline.py [out file]"""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from gansynth_amd import kernels, variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict

so = os.path.join(ROOT, "scripts", "probe", "libstamp.so")
if not os.path.exists(so):
    subprocess.run(["hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", os.path.join(ROOT, "scripts", "probe", "stamp.hip"), "-o", so], check=True)
lib = ctypes.CDLL(so)
lib.stamp.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
slots = torch.zeros(256, dtype=torch.int64, device="cuda")
names = []


def stamp(tag):
    if not torch.cuda.is_current_stream_capturing():
        return
    names.append(tag)
    lib.stamp(slots.data_ptr(), len(names) - 1, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


def wrap(obj, name, tag):
    fn = getattr(obj, name)

    def inner(*a, **kw):
        stamp(tag + " >")
        try:
            return fn(*a, **kw)
        finally:
            stamp(tag + " <")
    setattr(obj, name, inner)


dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pg = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
real_fn, fake_fn = bench.synthetic_inputs(8, 0, dtype) if hasattr(bench, "synthetic_inputs") else (None, None)
if real_fn is None:
    g = torch.Generator(device="cuda").manual_seed(1)
    lab = torch.nn.functional.one_hot(torch.randint(0, 61, (8,), device="cuda", generator=g), 61).to(dtype)
    img = torch.randn(8, 2, 128, 1024, device="cuda", generator=g).clamp(-1, 1).to(dtype).contiguous(memory_format=torch.channels_last)
    real_fn = lambda: (img, lab)
    fake_fn = lambda: torch.randn(8, 256, device="cuda", dtype=dtype)
hp = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
          discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
DIST = bool(os.environ.get("PT_DIST"))   # world size 1 on the library's RCCL communicator (GS_COMM_MARKER_US: stand-ins for the two all-reduces)
if DIST:
    import torch.distributed as dist
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % (29400 + os.getpid() % 500), rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = GANSynth(pg.generator, pg.discriminator, real_fn, fake_fn, None, hp, dtype=dtype, use_graphs=True, distributed=DIST)
K = kernels.get()
wrap(model, "_reduce_in_capture", "all-reduce")
wrap(model, "_d_losses_a", "D real fwd + R1 first-order")
wrap(model, "_run_before_fake", "Adam(G) pending")
wrap(model, "_d_losses_b", "D fake pass (G fwd, D fwd) + loss")
wrap(model, "_g_losses_a", "G.A (G fwd + mode-seeking first-order)")
wrap(model, "_g_losses_b", "G.B: D fwd on G(z) + loss")
wrap(model, "_apply_in_graph", "optimizer step in graph")
orig_flush = K.flush_wgrad_reductions


def flush(*a, **kw):
    tag = "early contraction" if kw.get("select") is not None else "final contraction"
    stamp(tag + " >")
    try:
        return orig_flush(*a, **kw)
    finally:
        stamp(tag + " <")
K.flush_wgrad_reductions = flush
orig_pb = model._part_b


def part_b(which, *a, **kw):
    stamp("part B of the %s run >" % which)
    try:
        return orig_pb(which, *a, **kw)
    finally:
        stamp("part B of the %s run <" % which)
model._part_b = part_b

for _ in range(6):
    model.train_step()
model.synchronize()
import time
t0 = time.perf_counter()
for _ in range(20):
    model.train_step()
model.synchronize()
ms = (time.perf_counter() - t0) / 20 * 1e3
vals = slots.cpu().tolist()[:len(names)]
base = min(v for v in vals if v)
out = open(sys.argv[1], "w") if len(sys.argv) > 1 else sys.stdout
if DIST:
    dist.destroy_process_group()
print("# %.3f ms per iteration with %d stamp nodes in the graph; us since the earliest stamp of the LAST replay (stamp kernels queue in their stream's order)" % (ms, len(names)), file=out)
for tag, v in sorted(zip(names, vals), key=lambda t: t[1]):
    print("%9.1f us  %s" % ((v - base) / 100.0, tag), file=out)
