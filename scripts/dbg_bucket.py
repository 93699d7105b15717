"""Data parallel at world size 1 (the library's RCCL communicator): a few full-size iterations with the discriminator's all-reduce in two steps
(GS_DP_BUCKET_D=1); GS_DEBUG_DP_BUCKET=1 prints the range the first message covers.  usage: GS_DP_BUCKET_D=1 GS_DEBUG_DP_BUCKET=1 python scripts/dbg_bucket.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29477", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from tests.test_model_gpu import _dp_trainer, R
from gansynth_amd import kernels
batches = [R.synthetic_batch(8, rank=i, image_shape=(2, 128, 1024)) for i in range(3)]
model = _dp_trainer(1.0, batches, full=True, dtype=torch.bfloat16, keep=False)
print("bucket_d_reduce", model.bucket_d_reduce, "split_final_flush", model.split_final_flush, "graph_allreduce", model._graph_allreduce)
for _ in range(3): model.train_step()
model.synchronize()
print("fused", bool(model._merged and model._merged.get("fused")), "expected tags", {k: len(v) for k, v in kernels.get()._expected.items()})
dist.destroy_process_group()
