"""Which masked launches of one eager iteration (full size, bf16) read sign bits and which read the activation values (debugging aid for the 1-bit
leaky-relu masks: kernels._mask_act).  usage: python scripts/dbg_mask_bits.py"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import kernels, variables

seen = collections.Counter()
orig = kernels._mask_act


def spy(mask, mask_act):
    r = orig(mask, mask_act)
    seen[(tuple(mask.shape), "bits" if r == 5 else "values(act %d)" % mask_act, mask.storage_offset(), mask.untyped_storage().nbytes() - mask.numel() * 2)] += 1
    return r


kernels._mask_act = spy
dtype = torch.bfloat16
lat, lab, real = R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs = False
model.real_input_fn = lambda: (cuda(real).to(dtype), cuda(lab).to(dtype))
model.fake_input_fn = lambda: cuda(lat).to(dtype)
gp, dp = opg.init_params(seed=0, bias_std=0.1)
model._build(cuda(lat).to(dtype), cuda(lab).to(dtype))
variables.default_store().load_state_dict({**gp, **dp})
model.train_step(); model.synchronize()
for k, v in sorted(seen.items()):
    print(v, k)
