"""Replays the forked generator-run graph several times on the same state and lists the kernel-layer operands / results that differ
between the replays (copies of every tensor argument and result are captured with the run: kernels._StreamGuard.hook)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_model_gpu import make, cuda, R
from gansynth_amd import variables, kernels

dtype = torch.bfloat16
lat, lab, real = R.synthetic_batch(8, rank=0, image_shape=(2, 128, 1024))
variables.set_default_store(variables.VariableStore(device="cuda"))
pg, opg, model = make(1.0, variables.default_store(), full=True, dtype=dtype)
model.use_graphs = True
gp, dp = opg.init_params(seed=0, bias_std=0.1)
lat, lab, real = cuda(lat).to(dtype), cuda(lab).to(dtype), cuda(real).to(dtype)
model._build(lat, lab)
variables.default_store().load_state_dict({**gp, **dp})
K = kernels.get()
which = os.environ.get("DBG_RUN", "g")
inputs = (lat, lab) if which == "g" else (lat, lab, real)
rec = []
keepalive = []
recording = [False]


def record_hook(inner, name):
    """kernels._StreamGuard.hook: copies of every operand and result of every kernel-layer call, captured WITH the run (which tensor differs
    between two replays?); DBG_KEEP: nothing a kernel touched is freed -- hence no block reused -- before `keepalive` is dropped."""
    def wrapper(*a, **kw):
        if not recording[0]:
            return inner(*a, **kw)
        stream = torch.cuda.current_stream()
        ins = [(i, t.data_ptr(), t.numel() * t.element_size(), t.clone()) for i, t in enumerate(list(a) + list(kw.values()))
               if isinstance(t, torch.Tensor) and t.is_cuda]
        out = inner(*a, **kw)
        if os.environ.get("DBG_KEEP"):
            keepalive.append((a, kw, out))
        outs = [(i, t.data_ptr(), t.numel() * t.element_size(), t.clone()) for i, t in enumerate(out if isinstance(out, (tuple, list)) else (out,))
                if isinstance(t, torch.Tensor) and t.is_cuda]
        rec.append((name, int(stream.cuda_stream), ins, outs))
        return out
    return wrapper


kernels._StreamGuard.hook = staticmethod(record_hook)
orig = model._forward_backward
def fb(w, *a):
    recording[0] = torch.cuda.is_current_stream_capturing()
    try:
        return orig(w, *a)
    finally:
        recording[0] = False
model._forward_backward = fb

def checksum(t):
    v = t.contiguous().reshape(-1).view(torch.uint8).to(torch.int64)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 251 + 1
    return int((v.flatten() * w).sum())

snaps = []
for rep in range(4):
    if os.environ.get('DBG_PERTURB'):
        lat = lat + 0.01 * rep
        inputs = (lat, lab) if which == 'g' else (lat, lab, real)
    model._run(which, *inputs)
    torch.cuda.synchronize()
    snaps.append([([checksum(t[3]) for t in ins], [checksum(t[3]) for t in outs]) for _, _, ins, outs in rec])
main_stream = rec[0][1]
print(len(rec), "recorded calls; branches", model.branches_opened, "; calls on the side stream:", sum(1 for r in rec if r[1] != main_stream))
for ra in range(4):
    for rb in range(ra + 1, 4):
        nd = sum(1 for (i0, o0), (i1, o1) in zip(snaps[ra], snaps[rb]) if i0 != i1 or o0 != o1)
        print("replay %d vs %d: %d calls with differing operands / results" % (ra, rb, nd))
for rep in range(1, 2):
    print("replay 0 vs %d" % rep)
    shown = 0
    for ci, ((name, st, ins, outs), (i0, o0), (i1, o1)) in enumerate(zip(rec, snaps[0], snaps[rep])):
        bi = [ins[j][0] for j in range(len(ins)) if i0[j] != i1[j]]
        bo = [outs[j][0] for j in range(len(outs)) if o0[j] != o1[j]]
        if bi or bo:
            print("   call #%d %-28s %s  operands differing %s  results differing %s" % (ci, name, "side" if st != main_stream else "main", bi, bo))
            if shown == 0:
                for j in bi:
                    idx, ptr, nb, _ = next(t for t in ins if t[0] == j)
                    print("      operand %d at %x (%d bytes); calls that touched an overlapping range before:" % (j, ptr, nb))
                    for cj in range(ci):
                        n2, st2, ins2, outs2 = rec[cj]
                        for kind, lst in (("in", ins2), ("out", outs2)):
                            for (i2, p2, nb2, _) in lst:
                                if p2 < ptr + nb and ptr < p2 + nb2:
                                    print("         #%d %-28s %s %s[%d] at %x (%d bytes)" % (cj, n2, "side" if st2 != main_stream else "main", kind, i2, p2, nb2))
            if shown == 0 and rep == 1:
                for (idx, ptr, nb, _) in ins:
                    print("      operand %d at %x (%d bytes); calls that touch an overlapping range:" % (idx, ptr, nb))
                    for cj in range(len(rec)):
                        n2, st2, ins2, outs2 = rec[cj]
                        for kind, lst in (("in", ins2), ("out", outs2)):
                            for (i2, p2, nb2, _) in lst:
                                if p2 < ptr + nb and ptr < p2 + nb2:
                                    print("         #%d %-28s %s %s[%d] at %x (%d bytes)" % (cj, n2, "side" if st2 != main_stream else "main", kind, i2, p2, nb2))
            if shown == 0:
                for j in bo:
                    idx, ptr, nb, _ = next(t for t in outs if t[0] == j)
                    print("      result %d at %x (%d bytes); calls that touch an overlapping range:" % (j, ptr, nb))
                    for cj in range(len(rec)):
                        n2, st2, ins2, outs2 = rec[cj]
                        for kind, lst in (("in", ins2), ("out", outs2)):
                            for (i2, p2, nb2, _) in lst:
                                if p2 < ptr + nb and ptr < p2 + nb2:
                                    print("         #%d %-28s %s %s[%d] at %x (%d bytes)" % (cj, n2, "side" if st2 != main_stream else "main", kind, i2, p2, nb2))
                print("      calls around it:")
                for cj in range(max(0, ci - 25), min(len(rec), ci + 6)):
                    n2, st2, ins2, outs2 = rec[cj]
                    print("         #%d %-28s %s  in %s  out %s" % (cj, n2, "side" if st2 != main_stream else "main",
                          ["%x+%d" % (p2, nb2) for _, p2, nb2, _ in ins2], ["%x+%d" % (p2, nb2) for _, p2, nb2, _ in outs2]))
            shown += 1
            if shown >= 6:
                break
