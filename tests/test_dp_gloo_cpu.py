"""CPU, world_size 2, gloo: the data-parallel path of models.GANSynth.

Each rank runs the trainer on its own shard (local batch 4 -- batch_stddev groups stay inside a rank) with
gradients all-reduced and averaged inside the Adam step.  Reference: one process that computes both shards'
gradients itself and averages them.  Kernel layer = the torch-CPU emulation (tests/cpu_kernels.py)."""
import os
import sys
import tempfile

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup(level=0.3):
    from gansynth_amd import kernels, variables
    from gansynth_amd.models import GANSynth
    from gansynth_amd.networks import PGGAN
    from gansynth_amd.utils import Dict
    from oracle import torch_ref as R
    from tests.cpu_kernels import CpuEmuKernels
    kernels.set_backend(CpuEmuKernels())
    variables.set_default_store(variables.VariableStore(device="cpu", seed=0))
    pg = PGGAN(min_resolution=[2, 16], max_resolution=[8, 64], min_channels=8, max_channels=16, growing_level=level)
    hyper = Dict(R.DEFAULT_HYPER)
    return pg, hyper, GANSynth


def _shard(rank):
    g = torch.Generator().manual_seed(100 + rank)
    lat = torch.randn(4, 16, generator=g)
    lab = torch.nn.functional.one_hot(torch.randint(0, 5, (4,), generator=g), 5).float()
    img = torch.randn(4, 2, 8, 64, generator=g).clamp(-1, 1)
    return lat, lab, img


def _init(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)


def _worker(rank, world, port, out_dir, bucket_bytes):
    _init(rank, world, port)
    pg, hyper, GANSynth = _setup()
    model = GANSynth(pg.generator, pg.discriminator, None, None, None, hyper, distributed=True, bucket_bytes=bucket_bytes, keep_gradients=True)
    lat, lab, img = _shard(rank)
    model.discriminator_step(lat, lab, img)
    nd = len(model.d_params.buckets)
    model.generator_step(lat, lab)
    torch.save({"d": model.d_params.flat.clone(), "g": model.g_params.flat.clone(), "step": model.global_step,
                "buckets": (nd, len(model.g_params.buckets)), "g_buckets": model.g_params.buckets,
                "d_grad": model.d_params.grad.clone(), "g_grad": model.g_params.grad.clone(),   # (the all-reduced SUMS: Adam does not touch them)
                "g_offsets": list(model.g_params._offsets)},
               os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,bucket_bytes", [(2, 8 << 20), (4, 16 << 10), (2, 4 << 10), (2, 8 << 10)])
def test_gloo_ranks_match_gradient_average(world, bucket_bytes):
    """world 2 with one bucket per network (the whole flat gradient) and world 4 with 16 KiB buckets: the bucketed path -- buckets
    of whole tensors in completion order, each all-reduce launched from inside the backward's tail as its last gradient lands
    (kernels.flush_wgrad_reductions per bucket), TF-Adam bucket by bucket behind its own all-reduce."""
    sys.path.insert(0, ROOT)
    port = 29500 + (os.getpid() % 2000) + world
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_worker, args=(world, port, d, bucket_bytes), nprocs=world, join=True)
        rs = [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world)]
    r0 = rs[0]
    for r in rs[1:]:
        assert torch.equal(r0["d"], r["d"]) and torch.equal(r0["g"], r["g"])  # replicas stay bit-identical
    assert r0["step"] == 1
    if bucket_bytes < (1 << 20):
        assert r0["buckets"][0] > 2 and r0["buckets"][1] > 2, r0["buckets"]
        gb = r0["g_buckets"]
        assert gb[0][0] > gb[-1][0]   # generator: completion order = reverse of the variable order (its backward ends at the embedding)
        cover = sorted(gb)
        if bucket_bytes < (16 << 10):   # some conv weight and its bias sit in different buckets
            where = {name: next(i for i, (a, b) in enumerate(gb) if a <= off < b) for off, _, name in r0["g_offsets"]}
            assert any(where[k] != where[k[:-len("weight")] + "bias"] for k in where if k.endswith("conv/weight")), where
        assert cover[0][0] == 0 and all(a[1] == b[0] for a, b in zip(cover, cover[1:])) and cover[-1][1] == r0["g"].numel()


    # single-process reference: average of the per-shard gradients, then one TF-Adam step
    from gansynth_amd import kernels, variables
    old_k, old_s = kernels._K, variables._default
    try:
        pg, hyper, GANSynth = _setup()
        model = GANSynth(pg.generator, pg.discriminator, None, None, None, hyper)
        lat0, lab0, img0 = _shard(0)
        model._build(lat0, lab0)
        import math
        for which in ("d", "g"):
            params = model.d_params if which == "d" else model.g_params
            other = model.g_params if which == "d" else model.d_params
            params.requires_grad_(True)
            other.requires_grad_(False)
            acc = torch.zeros_like(params.grad)
            for rank in range(world):
                lat, lab, img = _shard(rank)
                params.zero_grad()
                loss = (model.discriminator_losses(lat, lab, img) if which == "d" else model.generator_losses(lat, lab)).mean()
                loss.backward()
                acc += params.grad
            params.grad.copy_(acc)
            torch.testing.assert_close(r0[which + "_grad"], acc, rtol=1e-5, atol=1e-7)   # the reduced gradient itself, every element
            params.t += 1
            lr_t = 8e-4 * math.sqrt(1 - 0.99 ** params.t)
            kernels.get().adam_tf_step(params.flat, params.grad, params.m, params.v, lr_t, 0.0, 0.99, 1e-8, 1.0 / world)
        torch.testing.assert_close(r0["d"], model.d_params.flat, rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(r0["g"], model.g_params.flat, rtol=1e-5, atol=1e-6)
    finally:
        kernels._K, variables._default = old_k, old_s


# ---------------------------------------------------------------------------------------------- train(): resume and input exhaustion
def _finite_input(rank, n_batches):
    """real_input_fn with `n_batches` batches then StopIteration (an epoch-limited shard); images, not waveforms."""
    state = {"k": 0}

    def fn():
        if state["k"] >= n_batches:
            raise StopIteration
        state["k"] += 1
        _, lab, img = _shard(rank * 100 + state["k"])
        return img, lab

    fn.finite = True
    return fn


def _train_worker(rank, world, port, out_dir, model_dir, total_steps, batches):
    _init(rank, world, port)
    pg, hyper, GANSynth = _setup(level=lambda: 0.3)
    gen = torch.Generator().manual_seed(7 + rank)
    model = GANSynth(pg.generator, pg.discriminator, _finite_input(rank, batches[rank]), lambda: torch.randn(4, 16, generator=gen), None, hyper,
                     distributed=True, bucket_bytes=16 << 10)
    model.train(total_steps=total_steps, log=None, model_dir=model_dir, save_checkpoint_steps=0)
    torch.save({"d": model.d_params.flat.clone() if model.d_params is not None else None,
                "g": model.g_params.flat.clone() if model.g_params is not None else None, "step": model.global_step,
                "t": (model.d_params.t, model.g_params.t) if model.d_params is not None else None, "restored": model.restored_from,
                "files": sorted(os.listdir(model_dir))}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


def _run_train(world, model_dir, total_steps, batches):
    port = 31500 + (os.getpid() % 2000) + total_steps
    with tempfile.TemporaryDirectory() as d:
        mp.spawn(_train_worker, args=(world, port, d, model_dir, total_steps, batches), nprocs=world, join=True)
        return [torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(world)]


@pytest.mark.timeout(900)
def test_two_rank_resume_and_uneven_input():
    """(i) Every rank restores the SAME checkpoint (weights, Adam slots, optimizer steps, global_step) so that a resumed
    data-parallel run continues in one growing regime with identical replicas; only rank 0 writes.  (ii) Rank-local inputs that run
    dry at different steps stop every rank at the same iteration instead of hanging the others in an all-reduce."""
    sys.path.insert(0, ROOT)
    with tempfile.TemporaryDirectory() as model_dir:
        first = _run_train(2, model_dir, 2, [100, 100])
        assert [r["step"] for r in first] == [2, 2] and all(r["restored"] is None for r in first)
        assert first[0]["files"] == ["checkpoint", "checkpoints_keep_clock", "model.ckpt-2.safetensors"]   # written once, by rank 0
        again = _run_train(2, model_dir, 3, [100, 100])
        assert all(r["restored"] is not None and r["restored"].endswith("model.ckpt-2.safetensors") for r in again)
        assert [r["step"] for r in again] == [3, 3] and all(r["t"] == (3, 3) for r in again)
        assert torch.equal(again[0]["d"], again[1]["d"]) and torch.equal(again[0]["g"], again[1]["g"])
        assert not torch.equal(again[0]["d"], first[0]["d"])
        done = _run_train(2, model_dir, 3, [100, 100])                                     # a finished run resumes to zero further steps
        assert [r["step"] for r in done] == [3, 3] and torch.equal(done[0]["g"], again[0]["g"])
    with tempfile.TemporaryDirectory() as model_dir:
        # each iteration draws two batches (D run, G run): rank 0 can do 3 iterations, rank 1 only 2 -> both stop after 2
        uneven = _run_train(2, model_dir, 50, [6, 4])
        assert [r["step"] for r in uneven] == [2, 2]
        assert torch.equal(uneven[0]["d"], uneven[1]["d"]) and torch.equal(uneven[0]["g"], uneven[1]["g"])


# ---------------------------------------------------------------------------------------------- rank-local paths never communicate
def _pending_worker(rank, world, port, out_dir, model_dir):
    """The state machine of the in-graph pipelined step (models.GANSynth._train_step_pipelined: the generator's gradient is left
    UNREDUCED and its update pending until the next discriminator graph) emulated on CPU: train() with a checkpoint every step.  Every
    all-reduce a rank issues is logged; the sequences must be identical on the two ranks although only rank 0 saves."""
    _init(rank, world, port)
    pg, hyper, GANSynth = _setup(level=lambda: 0.3)
    gen = torch.Generator().manual_seed(7 + rank)

    class Pipelined(GANSynth):
        def train_step(self):
            real_images, labels, d_latents, g_latents, g_labels = self._next_inputs()
            self._ensure_built(d_latents, labels)
            d_loss = self.discriminator_step(d_latents, labels, real_images)   # (_run joins the pending generator update first)
            g_loss = self._run("g", g_latents, g_labels)                        # gradients only: no reduction, no update
            if self._pipe is None:
                self._pipe = {"g_pending": False, "g_unreduced": False, "key": None}
            self._pipe["g_pending"] = self._pipe["g_unreduced"] = True
            self.global_step += 1
            return d_loss, g_loss

    calls = []
    real_all_reduce = torch.distributed.all_reduce

    def logged(tensor, *a, **kw):
        calls.append(int(tensor.numel()))
        return real_all_reduce(tensor, *a, **kw)

    torch.distributed.all_reduce = logged
    model = Pipelined(pg.generator, pg.discriminator, _finite_input(rank, 100), lambda: torch.randn(4, 16, generator=gen), None, hyper,
                      distributed=True, bucket_bytes=8 << 20)
    model.train(total_steps=3, log=None, model_dir=model_dir, save_checkpoint_steps=1)
    assert not model.collective_pending()
    torch.save({"calls": calls, "g": model.g_params.flat.clone(), "d": model.d_params.flat.clone(), "t": (model.d_params.t, model.g_params.t),
                "files": sorted(os.listdir(model_dir))}, os.path.join(out_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(600)
def test_checkpoint_in_mid_training_does_not_communicate_from_rank_0_alone():
    """with the generator's all-reduce pending, checkpoint.save -> state_dict -> synchronize() on rank 0 alone issued
    a collective its peers never matched.  train() now joins the pending update on EVERY rank before a rank-local save and at its end."""
    sys.path.insert(0, ROOT)
    port = 33500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as d, tempfile.TemporaryDirectory() as model_dir:
        mp.spawn(_pending_worker, args=(2, port, d, model_dir), nprocs=2, join=True)
        r0, r1 = (torch.load(os.path.join(d, f"rank{r}.pt")) for r in range(2))
    assert r0["calls"] == r1["calls"] and len(r0["calls"]) >= 6, (r0["calls"], r1["calls"])   # same collectives, same order, both ranks
    assert torch.equal(r0["g"], r1["g"]) and torch.equal(r0["d"], r1["d"])
    assert r0["t"] == r1["t"] == (3, 3)                                                        # every pending update was applied
    assert [f for f in r0["files"] if f.startswith("model.ckpt-")] == ["model.ckpt-1.safetensors", "model.ckpt-2.safetensors", "model.ckpt-3.safetensors"]
