"""Which aten ops with a device kernel does one training iteration still run outside libgansynth_hip, from which line?
(TorchDispatchMode: sees the ops of the autograd engine's thread too, with the Python frames of custom Function.backward bodies.)"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from gansynth_amd import variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict
from torch.utils._python_dispatch import TorchDispatchMode

VIEWS = ("view", "reshape", "permute", "transpose", "slice", "select", "expand", "as_strided", "detach", "alias", "unsqueeze", "squeeze", "t.default",
         "empty", "_unsafe_view", "is_", "size", "stride", "unbind", "split", "narrow", "new_empty", "lift_fresh", "_local_scalar_dense", "empty_like", "empty_strided")


class Log(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.agg = collections.Counter()
        self.seq = []

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            shapes = [tuple(a.shape) for a in args if isinstance(a, torch.Tensor)]
            if any(isinstance(a, torch.Tensor) and a.is_cuda for a in args) or "fill" in name or "zeros" in name:
                st = [f for f in traceback.extract_stack() if "gansynth_amd" in f.filename]
                where = " <- ".join("%s:%d" % (os.path.basename(f.filename), f.lineno) for f in st[-3:]) if st else "(engine)"
                if not st:
                    node = torch._C._current_autograd_node() if hasattr(torch._C, "_current_autograd_node") else None
                    where = "(engine)" if node is None else "(engine) in " + node.name()
                self.agg[(name, str(shapes)[:70], where)] += 1
                self.seq.append((name, str(shapes)[:70], where))
        else:
            st = [f for f in traceback.extract_stack() if "gansynth_amd" in f.filename]
            if st:
                self.seq.append(("  .. " + name, "", " <- ".join("%s:%d(%s)" % (os.path.basename(f.filename), f.lineno, f.name) for f in st[-2:])))
        return func(*args, **(kwargs or {}))


dtype = torch.bfloat16
variables.set_default_store(variables.VariableStore(device="cuda", seed=0))
pggan = PGGAN(min_resolution=[2, 16], max_resolution=[128, 1024], min_channels=32, max_channels=256, growing_level=1.0)
hyper = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
             discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
pool = bench.synthetic_pool(8, 0, dtype)
model = GANSynth(pggan.generator, pggan.discriminator, lambda: (pool[0][2], pool[0][1]), lambda: pool[0][0], None, hyper, dtype=dtype, use_graphs=False)
for _ in range(2):
    model.train_step()
torch.cuda.synchronize()
log = Log()
with log:
    model.train_step()
torch.cuda.synchronize()
for (name, shapes, where), n in sorted(log.agg.items(), key=lambda kv: (kv[0][2], kv[0][0])):
    print("%3d x %-28s %-72s %s" % (n, name.replace("aten.", ""), shapes, where))
print("total", sum(log.agg.values()))
# what runs right behind each op issued by the autograd engine itself (gradient accumulation of a tensor with several consumers)?
for i, ent in enumerate(log.seq):
    if ent[2].startswith("(engine)") and not ent[0].startswith("  .."):
        print("ENGINE", ent[0], ent[1])
        print("   ", ent[2])
        for e in log.seq[max(0, i - 5):i]:
            print("      before:", e)
        for e in log.seq[i + 1:i + 6]:
            print("      after: ", e)

