"""World size 1 with the library's RCCL communicator: where do the two gradient all-reduces of an iteration sit?  RCCL short-cuts a one-rank
all-reduce to nothing, so GS_COMM_MARKER_US=<n> puts a one-block kernel that holds its stream for n us in its place; the iteration is timed
with markers of 0 and n us in every schedule named on the command line ("ENV=1,ENV2=0" per schedule, "-" = defaults).
usage (GPU box): python scripts/dp_marker_check.py 300 - GS_FAKE_FIRST=0 GS_HOOK_AFTER_BACKWARD=1 ..."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%%d" %% (29400 + os.getpid() %% 500), rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tests.test_model_gpu import _dp_trainer, R
batches = [R.synthetic_batch(8, rank=i, image_shape=(2, 128, 1024)) for i in range(3)]
model = _dp_trainer(1.0, batches, full=True, dtype=torch.bfloat16, keep=False)
for _ in range(4):
    model.train_step()
model.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(8):
        model.train_step()
    model.synchronize()
    best = min(best, (time.perf_counter() - t0) / 8 * 1e3)
print("MS %%.4f fused=%%s" %% (best, bool(model._merged and model._merged.get("fused"))))
dist.destroy_process_group()
''' % ROOT

us = sys.argv[1]
for sched in sys.argv[2:]:
    env_extra = {} if sched == "-" else dict(kv.split("=") for kv in sched.split(","))
    out = {}
    for marker in ("0", us):
        env = dict(os.environ, GS_COMM_MARKER_US=marker, **env_extra)
        res = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, cwd=ROOT)
        line = [l for l in res.stdout.splitlines() if l.startswith("MS ")]
        out[marker] = line[-1] if line else "FAILED: " + (res.stderr or res.stdout)[-300:]
    try:
        a, b = float(out["0"].split()[1]), float(out[us].split()[1])
        print("[%s] marker 0: %.3f ms   marker %s us x 2: %.3f ms   -> the two stand-ins add %.3f ms  (%s)" % (sched, a, us, b, b - a, out["0"].split()[2]))
    except Exception:
        print("[%s]" % sched, out)
