"""GANSynth trainer step on the HIP path (reference models.py:8-108, train loop :189-194).

Same constructor as the reference's GANSynth (generator / discriminator callables, input fns,
spectral_params, hyper_params).  One iteration = a discriminator update then a generator update,
each on its own fresh batch, exactly like the two session.run calls of models.py:191-192:

  D run: L_D = mean(softplus(-r) + softplus(f) + w_r1 * sum((d sum(r) / d x_real)^2))   :39-49,65
  G run: L_G = mean(softplus(-f) + w_ms / (sum((d sum(G(z)) / d z)^2) + 1e-6))            :57-64
  both with tf.train.AdamOptimizer (TF form), only the G run bumps global_step            :67-89

Parameters, gradients and Adam slots of each network live in one flat fp32 buffer (one fused
Adam launch, one all-reduce payload).  Data parallelism (new -- the reference is single GPU):
one process per GPU, gradients summed with torch.distributed all_reduce (backend "nccl" = RCCL
over xGMI) and averaged inside the Adam kernel.
"""
import contextlib
import math
from collections import OrderedDict

import torch
import torch.nn.functional as TF

from . import config
from . import functional as F
from . import kernels
from . import spectral_ops
from . import variables

_PAD = 64  # floats; keeps every parameter view 256-byte aligned inside the flat buffer


_DEFER_REDUCTIONS = not config.flag("GS_NO_DEFERRED_REDUCE")   # A/B switches for measurements
_FUSED_LOSSES = not config.flag("GS_NO_FUSED_LOSSES")
_BATCH_D_TAIL = not config.flag("GS_NO_D_TAIL_BATCH")   # A/B switch: real + fake through the discriminator's tail as one batch
_PIPELINE = config.flag("GS_PIPELINE")   # opt-in, see GANSynth.pipeline
_PIPE_SIDE = {"0": False, "1": True}.get(config.value("GS_PIPE_SIDE", ""))
# The all-reduce beside part A of the other run (forked graph branch, four graphs per iteration: round 4's default) is opt-in since round 5:
# the two-graph form with the collective as the LAST node of each run's graph keeps the compute branches of section 6.5 (a four-graph
# iteration with branches would be launch-bound) -- 5.16 against 5.68 ms at world size 1, i.e. the overlapped form has to hide more than
# half a millisecond of all-reduce to break even.
_OVERLAP_REDUCE = config.flag("GS_OVERLAP_REDUCE") and not config.flag("GS_NO_OVERLAP_REDUCE")
_GRAPH_ALLREDUCE = not config.flag("GS_NO_GRAPH_ALLREDUCE")   # A/B switch: the gradient all-reduce as a node of the captured graph
_FORK = not config.flag("GS_NO_FORK")   # A/B switch: independent sub-passes of a run on a forked branch of its hipGraph (GANSynth._branch)
LEVEL_STREAMS = int(config.value("GS_LEVEL_STREAMS", "128"))   # see GANSynth._leveled_queues
EARLY_FLUSH_DIVS = [int(d) for d in config.value("GS_EARLY_FLUSH_DIV", "16").split(",")]   # a layer is "large" from 1/DIV of the full resolution's pixels (several: one early contraction each)
EARLY_FLUSH_CUS = int(config.value("GS_EARLY_FLUSH_CUS", "192"))   # see GANSynth._early_flush
_SUB_RUNS = config.flag("GS_SUB_RUNS")   # opt-in (measured slower, see _d_sub_runs): the discriminator run as two independent sub-runs
_FAKE_FIRST = config.flag("GS_FAKE_FIRST")   # opt-in, see _d_fake_first
_FORK_EAGER = config.flag("GS_FORK_EAGER")   # tests: the same branches with eager launches (a second stream, event hops)


def _capture_mode(with_collective, forked=False):
    """Keyword arguments of torch.cuda.graph for a capture that contains an RCCL collective: the communicator's helper threads may call
    the HIP runtime while this thread captures (proxy progress, registration), which the default "global" capture mode turns into a capture
    error on THEIR call -- captures with a collective inside run "thread_local" (only this thread's calls are checked), as captured NCCL
    work is run elsewhere.  With forked branches in the same capture (GANSynth._branch: autograd's device thread then records and waits on
    events between two captured streams) a thread_local capture replayed into a segmentation fault on this stack (ROCm 7.0.2, RCCL 2.26.6,
    one rank; "global" and "relaxed" captures of the same run replay fine): those captures are "relaxed" (no thread's calls are checked).
    Everything else keeps the strict default."""
    forced = config.value("GS_CAPTURE_MODE")   # (debugging)
    if forced:
        return {"capture_error_mode": forced}
    if not with_collective:
        return {}
    return {"capture_error_mode": "relaxed" if forked else "thread_local"}


def _hw_queues_allow_branches():
    """Graphs with parallel branches need the HIP runtime's default of four hardware queues (GPU_MAX_HW_QUEUES): with two, the streams an
    executable graph makes for its branches cannot all miss the launch stream's queue and hip::Graph::UpdateStreams walks off its list
    (see GANSynth._leveled_queues; measured: GPU_MAX_HW_QUEUES=2 crashes at the first replay, =8 runs at 7.8 ms instead of 5.2)."""
    n = __import__("os").environ.get("GPU_MAX_HW_QUEUES")
    if n is None:
        return True
    try:
        ok = int(n) == 4
    except ValueError:
        ok = False
    if not ok:
        import sys
        print("gansynth_amd.models: GPU_MAX_HW_QUEUES=%s: the runs' graphs are captured without parallel branches (they need the default, 4)" % n,
              file=sys.stderr, flush=True)
    return ok


# The forked schedule was debugged on ONE build of the HIP runtime (ROCm 7.0.2: the stream-list defect of hipGraphLaunch and the workaround
# for it, GANSynth._leveled_queues).  On that build the branches are used as they are; on any other the first trainer of a process that is about
# to capture a forked graph runs a reduced forked iteration in a CHILD process first -- a defect of this kind is a segmentation fault inside
# the runtime, which no exception handler of this process would see -- and a child that dies switches the branches off for the process (and
# for its children: GS_FORK_PROBED), with a line on stderr.  GS_FORK_PROBE=always / never overrides.
_VALIDATED_HIP = ("7.0.51831",)
_FORK_PROBE_RESULT = []
_FORK_PROBE = r"""
import sys, torch
sys.path.insert(0, %r)
from gansynth_amd import variables
from gansynth_amd.models import GANSynth
from gansynth_amd.networks import PGGAN
from gansynth_amd.utils import Dict
variables.set_default_store(variables.VariableStore(device="cuda"))
pg = PGGAN(min_resolution=[2, 16], max_resolution=[16, 128], min_channels=32, max_channels=64, growing_level=1.0)
hp = Dict(generator_learning_rate=8e-4, generator_beta1=0.0, generator_beta2=0.99, discriminator_learning_rate=8e-4, discriminator_beta1=0.0,
          discriminator_beta2=0.99, mode_seeking_loss_weight=0.1, real_gradient_penalty_weight=5.0, fake_gradient_penalty_weight=0.0)
dt = torch.bfloat16
lab = torch.nn.functional.one_hot(torch.arange(4) %% 61, 61).to("cuda", dt)
real = lambda: (torch.randn(4, 2, 16, 128, device="cuda").clamp(-1, 1).contiguous(memory_format=torch.channels_last).to(dt), lab)
model = GANSynth(pg.generator, pg.discriminator, real, lambda: torch.randn(4, 256, device="cuda", dtype=dt), None, hp, dtype=dt, use_graphs=True)
assert model.fork
for _ in range(4):
    model.train_step()
model.synchronize()
assert model.branches_opened > 0 and bool(torch.isfinite(model.g_params.flat).all())
print("forked replay ok")
"""


def _forked_replay_ok():
    if _FORK_PROBE_RESULT:
        return _FORK_PROBE_RESULT[0]
    import os
    import sys
    mode = config.value("GS_FORK_PROBE", "auto")
    hip = getattr(torch.version, "hip", None) or ""
    ok = True
    if config.value("GS_FORK_PROBED") in ("ok", "died"):
        ok = os.environ["GS_FORK_PROBED"] == "ok"
    elif mode == "never" or (mode != "always" and any(hip.startswith(v) for v in _VALIDATED_HIP)):
        ok = True
    else:
        import subprocess
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        try:
            res = subprocess.run([sys.executable, "-c", _FORK_PROBE % root], env=dict(os.environ, GS_FORK_PROBE="never"), capture_output=True, text=True,
                                 timeout=float(config.value("GS_FORK_PROBE_TIMEOUT_S", "600")))
            ok = res.returncode == 0 and "forked replay ok" in res.stdout
            why = "rc %s: %s" % (res.returncode, (res.stderr or res.stdout).strip().splitlines()[-1:] or "")
        except subprocess.TimeoutExpired:
            ok, why = False, "no result in time"
        os.environ["GS_FORK_PROBED"] = "ok" if ok else "died"
        if not ok:
            print("gansynth_amd.models: a forked hipGraph replay did not survive its probe on HIP %s (%s): the runs' graphs are captured without "
                  "parallel branches in this process" % (hip or "?", why), file=sys.stderr, flush=True)
    _FORK_PROBE_RESULT.append(ok)
    return ok


def _copy_inputs(dsts, srcs):
    """A run's inputs into the static buffers its graph reads: ONE multi-tensor launch where the tensors allow it (same device, dtype and
    strides pairwise) instead of a ~5 us copy kernel per input in front of every replay."""
    dsts, srcs = list(dsts), list(srcs)
    pairs = [(d, s) for d, s in zip(dsts, srcs) if d.data_ptr() != s.data_ptr()]
    if not pairs:
        return
    # (only the SMALL inputs share a launch: the multi-tensor kernel moves a 4 MB image batch on 34 blocks -- 21 us against 5 for its own copy)
    small = [(d, s) for d, s in pairs if d.numel() * d.element_size() <= (256 << 10)]
    if len(small) > 1 and all(d.is_cuda and s.is_cuda and d.dtype == s.dtype == small[0][0].dtype and d.stride() == s.stride() for d, s in small):
        torch._foreach_copy_([d for d, _ in small], [s for _, s in small])
        pairs = [(d, s) for d, s in pairs if d.numel() * d.element_size() > (256 << 10)]
    for d, s in pairs:
        d.copy_(s)


class _quiet_gc(object):
    """Collect garbage NOW and keep the cyclic collector off while a hipGraph is being captured: a collection in the middle of a
    capture may destroy an old CUDAGraph / event of an earlier trainer (a destructor that is illegal during capture: the process
    aborts).  torch.cuda.graph no longer collects on entry by itself."""

    def __enter__(self):
        import gc
        self._gc = gc
        gc.collect()
        self._was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        if self._was:
            self._gc.enable()
        return False


class _FlatParams(object):
    """All trainable variables of one scope re-homed into one flat fp32 buffer (+grad, m, v)."""

    def __init__(self, named_params):
        self.named = OrderedDict(named_params)
        sizes = [p.numel() for p in self.named.values()]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += (n + _PAD - 1) // _PAD * _PAD
        dev = next(iter(self.named.values())).device
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros_like(self.flat)
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.numel = sum(sizes)
        for (name, p), off, n in zip(self.named.items(), offs, sizes):
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            p.grad = self.grad[off:off + n].view(p.shape)
        self.t = 0
        self.grad_clean = True   # the flat gradient is all zeros (fresh buffer / cleared by the optimizer step)
        self.buckets = [(0, total)]
        self._offsets = list(zip(offs, sizes, self.named.keys()))

    def make_buckets(self, max_floats, reverse=False):
        """Split the flat buffer into contiguous ranges of whole tensors, each <= max_floats unless a single tensor is larger
        (the generator's 4.2 M-element dense weight gets a bucket of its own), ordered as the backward pass completes them:
        `reverse` for the generator (its backward ends at the first variables), natural order for the discriminator."""
        total = self.flat.numel()
        starts = [o for o, _, _ in self._offsets] + [total]
        buckets, a = [], 0
        for i in range(len(self._offsets)):
            nxt = starts[i + 1]
            if nxt - a > max_floats and starts[i] > a:      # closing before this tensor keeps the bucket under the limit
                buckets.append((a, starts[i]))
                a = starts[i]
            if nxt - a >= max_floats:
                buckets.append((a, nxt))
                a = nxt
        if a < total:
            buckets.append((a, total))
        self.buckets = buckets[::-1] if reverse else buckets
        return self.buckets

    def bucket_of(self, ptr):
        """Index (in completion order) of the bucket holding the gradient element at device address `ptr`, or None."""
        off = (ptr - self.grad.data_ptr()) // 4
        if 0 <= off < self.grad.numel():
            for i, (a, b) in enumerate(self.buckets):
                if a <= off < b:
                    return i
        return None

    def requires_grad_(self, flag):
        for p in self.named.values():
            p.requires_grad_(flag)

    def zero_grad(self):
        self.grad.zero_()
        self.grad_clean = False   # (about to be accumulated into)
        for p in self.named.values():
            if p.grad is None:
                raise RuntimeError("parameter lost its flat gradient view")

    def begin_run(self):
        """Gradients of a run are accumulated in place from zero.  The previous optimizer step may have left the buffer cleared
        (gs_adam_tf_step_zero_grad, `grad_clean`): then there is no fill pass."""
        if self.grad_clean:
            self.grad_clean = False
            for p in self.named.values():
                if p.grad is None:
                    raise RuntimeError("parameter lost its flat gradient view")
        else:
            self.zero_grad()


class GANSynth(object):

    def __init__(self, generator, discriminator, real_input_fn, fake_input_fn, spectral_params, hyper_params,
                 dtype=torch.float32, store=None, distributed=False, use_graphs=False, bucket_bytes=None, keep_gradients=False):
        self.generator, self.discriminator = generator, discriminator
        self.real_input_fn, self.fake_input_fn = real_input_fn, fake_input_fn
        self.spectral_params, self.hyper_params = spectral_params, hyper_params
        self.dtype = dtype
        self.store = store if store is not None else variables.default_store()
        self.distributed = bool(distributed)
        self.world = torch.distributed.get_world_size() if self.distributed else 1
        self.rank = torch.distributed.get_rank() if self.distributed else 0
        self.bucket_bytes = None if bucket_bytes is None else int(bucket_bytes)   # gradient all-reduce granularity (None: see _build)
        # False: the optimizer step clears the flat gradient it consumed (one pass less per run: the next run accumulates from zero);
        # True: `p.grad` of every variable still holds the run's (all-reduced, unscaled) gradient after the step -- tests, inspection
        self.keep_gradients = bool(keep_gradients)
        self._inflight = None                   # (params, [(bucket, work)]) all-reduces launched during the eager backward's tail
        self._comm = None                       # comm.RcclComm: the gradient all-reduce on the backward's own stream (HIP + nccl only)
        self._peeked = None                     # a batch fetched ahead of the first step (train: eager build / restore)
        self._run_reduced = False               # the last _run replayed a graph that contains its own gradient all-reduce
        self._captured_reduce = False
        self._graph_allreduce = _GRAPH_ALLREDUCE   # cleared for the life of the model if a capture with the collective inside fails
        self.global_step = 0
        self.g_params = None
        self.d_params = None
        self.generator_loss = None
        self.discriminator_loss = None
        # hipGraph replay of the forward+backward of each run (launch-bound otherwise: ~600 kernels per run).
        # Only valid while the network structure and every by-value kernel scalar are step-invariant, i.e. in the
        # fully grown regime (no fade coefficient); the optimizer update and the all-reduce stay outside the graph.
        self.use_graphs = bool(use_graphs)
        self._graphs = {}
        self._restore_from, self.restored_from = None, None
        self._graph_key, self._lerp = None, None
        # Pipelined iteration (opt-in: pipeline=True / GS_PIPELINE=1; train_step with graphs): every run is captured as two
        # graphs (own-network part A, the rest B) so that the gradient all-reduce of one network can run on a side stream under
        # part A of the other network's run, which needs none of it.  Off by default: on this ROCm stack a cross-stream event
        # hop between hipGraph replays costs 0.25-0.75 ms by itself (scripts/cross_stream_cost.py; the one-GPU step loses 6 %
        # with the side stream on and nothing to hide), about what an 8-GPU all-reduce of these 26 / 34 MiB buffers takes.
        self.pipeline = _PIPELINE
        self.pipe_side = None     # None: side stream iff data-parallel (see _train_step_pipelined)
        self._pipe = None
        # Data parallel on our own RCCL communicator with graphs: train_step() runs the pipelined iteration with each gradient
        # all-reduce as a FORKED BRANCH inside the other run's part-A graph (off the critical path; see "pipelined iteration").
        self.overlap_reduce = _OVERLAP_REDUCE
        self._pipe_capture = False
        self._warming_up = False
        # Forked branches inside a run's hipGraph (see _branch): a run is a chain of ~370 kernels of which ~150 are few-block launches of
        # the <= 8x64 levels -- 200 CUs idle while they run -- and it holds sub-passes that do not depend on each other.
        # (data parallel: off in the PIPELINED iteration unless GS_FORK_DIST=1 -- a graph with parallel branches costs the host 1.8 ms per
        #  replay instead of 0.06 (scripts/replay_host_time.py), and that iteration replays FOUR graphs: it would be launch-bound; decided
        #  in _build, when the transport is known.  The two-graph data-parallel forms keep their branches.)
        self.fork = _FORK and _hw_queues_allow_branches()
        self.fork_eager = _FORK_EAGER
        self.fork_marks = not config.flag("GS_NO_FORK_MARKS")   # (debugging: branches start where they are opened)
        self._side = None
        self._side2 = None        # the stream of a whole sub-run beside another run (_train_step_merged)
        self.bucket_d_reduce = config.flag("GS_DP_BUCKET_D")   # (data parallel, captured discriminator run, opt-in: _arm_first_bucket)
        self.first_bucket = None  # (the range of the flat gradient the first message of the last captured discriminator run covers)
        self._split_at = None     # (see _arm_first_bucket)
        self._first_bucket_stream = None
        self._side3 = None
        self._nodes_on_side2 = False
        self._origin = None
        self._after_loss = None
        self.merge_runs = not config.flag("GS_NO_MERGED_RUNS")   # A/B switch: see _train_step_merged
        self._merged = None
        # The WHOLE iteration as one hipGraph with both optimizer steps inside (see "one graph per iteration" above _capture_merged)
        self.fuse_iteration = not config.flag("GS_NO_FUSED_ITERATION")
        self._opt_scalars = None      # functional.DeviceScalars: [lr_t of the discriminator's step, lr_t of the generator's PENDING step | < 0]
        self._before_fake = None      # hook: issued on the fake pass's stream right before the generator's forward of a discriminator run
        self._in_sub_runs = False
        self._early_on_side2 = False
        self._fake_logits_early = None
        self.fake_first = _FAKE_FIRST   # (see _d_fake_first)
        self._g_ready = None          # event: the generator's weights (and prepared operands) of this iteration are final on the fake pass's stream
        self.sub_runs = _SUB_RUNS
        self.split_g_loss = not config.flag("GS_NO_SPLIT_G_LOSS")   # A/B switch, see _g_losses_b
        self.split_final_flush = not config.flag("GS_NO_SPLIT_FINAL_FLUSH")   # A/B switch, see kernels.HipKernels._flush_groups
        self._g_pending = None        # lr_t of a generator step whose gradient is in the flat buffer and whose update has not run yet
        self._marks = {}
        self._serial_run = False
        self._branched = False
        self.branches_opened = 0   # (tests / bench: how many branches the last captures opened)
        self.early_flush = not config.flag("GS_NO_EARLY_FLUSH")   # A/B switch: see _early_flush
        self.batch_d_tail = None   # None: the discriminator's tail over [real; fake] as one batch unless the runs fork (see _batched_tail)
        self.early_flush_always = False   # (tests: the same flush points without branches -- in place, on the one stream)
        self.early_flushes = 0
        self._early_in_run = 0

    # ------------------------------------------------------------------------ forked branches
    # A run holds sub-passes that do not depend on each other:
    #   G run: D(G(z)) forward                        ||  the mode-seeking first-order pass d sum(G(z)) / dz     (both need G(z) only)
    #          backward through D down to d L / d G(z) ||  the second-order pass of the mode-seeking term        (both need the loss head only)
    #   D run: G(z) forward (no grad)                  ||  D's trunk on the real batch
    # Each side has its own latency-bound stretch (the <= 8x64 levels: 64-block launches of ~10 us on a 256-CU chip) that the other side's
    # full-chip convs can fill.  Inside a stream capture a second stream is free -- fork and join become edges of the hipGraph, there is no
    # event hop at replay -- so the second member of each pair runs on a side stream THERE (and only there: between eager launches an event
    # hop costs more than the overlap gains).  autograd runs a node's backward on the stream its forward ran on and orders the streams with
    # events (captured as edges too), so putting D's forward on the side stream is what puts D's backward beside the second-order pass.
    # The host-side launch ORDER is the same with and without branches (the engine's ready queue does not look at streams): the same kernels
    # on the same operands, hence bit-identical parameters -- tests/test_model_gpu.py::test_forked_branches_change_nothing_but_the_schedule.
    # Memory: torch's caching allocator hands a freed block back to the stream that allocated it, so a tensor read on the other stream must
    # not be recycled under that read: kernels.HipKernels.stream_guard() marks every tensor argument of every kernel-layer call with the
    # stream it is used on (record_stream; inside a capture such a block is simply not reused before the capture ends).
    @contextlib.contextmanager
    def _leveled_queues(self):
        """Around a capture whose graph may hold parallel branches: LEVEL_STREAMS throw-away streams exist while the graph is instantiated
        (torch does that when the capture ends), so that the streams the HIP runtime makes for the branches land on different hardware
        queues -- see gs_streams_create in include/gansynth_hip.h for the runtime defect this keeps hipGraphLaunch away from."""
        # (also without branches of our own: the data-parallel graphs fork for their all-reduce)
        K = kernels.get() if ((self.fork or self.distributed) and torch.cuda.is_available()) else None
        if K is None or not hasattr(K, "lib") or LEVEL_STREAMS <= 0:
            yield
            return
        import ctypes
        from . import _lib
        handles = (ctypes.c_void_p * LEVEL_STREAMS)()
        ptr = ctypes.cast(handles, ctypes.POINTER(ctypes.c_void_p))
        t0 = __import__("time").perf_counter()
        try:
            _lib.check(K.lib.gs_streams_create(LEVEL_STREAMS, ptr), "gs_streams_create")   # (on failure the ones made so far are in `handles`)
            self.level_seconds = getattr(self, "level_seconds", 0.0) + __import__("time").perf_counter() - t0
            yield
        finally:
            _lib.check(K.lib.gs_streams_destroy(LEVEL_STREAMS, ptr), "gs_streams_destroy")

    def _check_fork_runtime(self):
        """Before the first capture that may hold parallel branches (see _forked_replay_ok)."""
        if self.fork and torch.cuda.is_available() and not _forked_replay_ok():
            self.fork = False

    def _forking(self):
        if not self.fork or not torch.cuda.is_available() or not hasattr(kernels.get(), "stream_guard"):
            return False
        return self._capturing() or (self.fork_eager and not self._warming_up)

    def _fork_mark(self, tag):
        """Remember this point of the current stream: a branch opened later IN THE SAME capture starts from here, not from the stream's end
        (serial runs only: the two parts of a pipelined run are two graphs, and an event of one capture cannot be waited on in another)."""
        self._marks.pop(tag, None)
        if self._serial_run and self._forking() and self.fork_marks:
            ev = torch.cuda.Event()
            ev.record()
            self._marks[tag] = ev

    def _second_stream(self, which, avoid):
        """A pooled stream for a branch that is none of `avoid` (torch.cuda.Stream() hands out 32 pooled streams round-robin: after enough
        captures -- every one takes a warm-up stream -- the next one IS the stream being captured: a branch that waits for itself)."""
        cur = getattr(self, which)
        taken = {a.cuda_stream for a in avoid if a is not None}
        if cur is None or cur.device != avoid[0].device or cur.cuda_stream in taken:
            for _ in range(64):
                cur = torch.cuda.Stream(device=avoid[0].device)
                if cur.cuda_stream not in taken:
                    break
            else:
                raise RuntimeError("no further stream for the forked branches of a captured run")
            setattr(self, which, cur)
        return cur

    @contextlib.contextmanager
    def _branch(self, tag=None, join=True):
        """`with self._branch(tag):` -- the launches inside go on the side stream, which starts at the mark `tag` (or here) and which the
        current stream waits for at the end of the block.  A no-op outside a capture."""
        ev = self._marks.pop(tag, None) if tag is not None else None
        if not self._forking():
            yield
            return
        main = torch.cuda.current_stream()
        self._second_stream("_side", [main, self._side2])
        side = self._side
        if ev is not None:
            side.wait_event(ev)
        else:
            side.wait_stream(main)
        self.branches_opened += 1
        self._branched = True
        with torch.cuda.stream(side):
            yield
        if join:   # (else: at the end of the run, _part_b)
            main.wait_stream(side)

    def _early_flush(self, select, then=None):
        """kernels.HipKernels.early_flush_rule: the weight gradients of the full-chip levels recorded so far are contracted NOW -- on the
        branch when the run is being captured (joined at the end of the run), else in place: the same launches on the same operands either
        way.  (Called from inside a backward node, i.e. on autograd's device thread, under that node's stream.)"""
        K = kernels.get()
        self.early_flushes += 1
        was = K.lib.gs_wgrad_cu_cap(EARLY_FLUSH_CUS) if hasattr(K, "lib") else 0   # (the chain beside it needs somewhere to land)
        cur = torch.cuda.current_stream().cuda_stream if torch.cuda.is_available() else None
        on_branch = self._side is not None and cur == self._side.cuda_stream
        # A node of the merged iteration's generator part runs on ITS stream: the contraction then goes to the run's own stream (idle while the
        # backward walks that one), never to the other branch -- two non-origin streams of a capture that wait on each other's events become each
        # other's parent in hip::Stream's bookkeeping and hip::Stream::EndCapture recurses until the stack runs out (ROCm 7.0.2).
        on_side2 = self._side2 is not None and cur == self._side2.cuda_stream and self._origin is not None and self._forking()
        try:
            if on_side2:
                self._origin.wait_stream(self._side2)
                with torch.cuda.stream(self._origin):
                    K.flush_wgrad_reductions(select=select)
                    if then is not None:
                        then()
                return
            if self._in_sub_runs and not on_branch and self._forking():
                # Sub-runs (_d_sub_runs): the contraction goes to the THIRD stream, behind part A of the generator run (issued between the
                # two sub-runs, _part_b), not behind the fake sub-run on the branch: that chain opens with the generator's all-reduce when the
                # job is data parallel, and whatever is queued behind it inherits the collective's time (the final contraction waits for
                # this one, _join_branches)
                main = torch.cuda.current_stream()
                third = self._second_stream("_side2", [main, self._side])
                third.wait_stream(main)
                if self._side is not None:
                    third.wait_stream(self._side)   # (the fake sub-run's pairs were recorded THERE: issued, not necessarily written yet)
                self._early_on_side2 = True
                self.branches_opened += 1
                with torch.cuda.stream(third):
                    K.flush_wgrad_reductions(select=select)
                    if then is not None:
                        then()
                return
            with (contextlib.nullcontext() if on_branch else self._branch(join=False)):   # (a node of the branch itself: in place)
                K.flush_wgrad_reductions(select=select)
                if then is not None:
                    then()
        finally:
            if hasattr(K, "lib"):
                K.lib.gs_wgrad_cu_cap(was)

    def _large_layer_pixels(self):
        owner = getattr(self.generator, "__self__", None)
        if owner is None or not hasattr(owner, "resolution"):
            return None
        full = int(owner.resolution(owner.max_depth).prod())
        return [max(1, full // d) for d in EARLY_FLUSH_DIVS]   # (16: the three levels at the top of the pyramid)

    def _stream_guard(self):
        K = kernels.get()
        if self.fork and hasattr(K, "stream_guard") and torch.cuda.is_available():
            return K.stream_guard()
        return contextlib.nullcontext()

    # ----------------------------------------------------------------------------- build
    def _build(self, latents, labels):
        """Create every variable (the reference's graph owns all of them from step 0), then flatten."""
        owner = getattr(self.generator, "__self__", None)
        with torch.no_grad():
            if owner is not None and hasattr(owner, "_g_variables"):
                with variables.variable_scope("generator"):
                    owner._g_variables(latents.shape[1], labels.shape[1])
                with variables.variable_scope("discriminator"):
                    owner._d_variables(labels.shape[1])
            else:
                images = self.generator(latents, labels)
                self.discriminator(images, labels)
        self.g_params = _FlatParams(self.store.trainable_variables("generator"))
        self.d_params = _FlatParams(self.store.trainable_variables("discriminator"))
        if self.distributed:  # identical weights on every rank
            from . import comm
            if self._comm is None and not config.flag("GS_TORCH_COLLECTIVES"):
                self._comm = comm.create(self.g_params.flat.device)   # RCCL on the backward's own stream (None on CPU / gloo)
            if self._comm is not None:
                self._comm.broadcast_(self.g_params.flat, 0)
                self._comm.broadcast_(self.d_params.flat, 0)
            else:
                torch.distributed.broadcast(self.g_params.flat, 0)
                torch.distributed.broadcast(self.d_params.flat, 0)
            if self.bucket_bytes is None:
                # same-stream RCCL: collectives and updates are serial on the one stream whatever the granularity, so ONE
                # all-reduce per network (fewest launches); torch.distributed's own collectives run on the communicator's stream
                # and do overlap the per-bucket updates: 8 MiB buckets there
                self.bucket_bytes = (64 << 20) if self._comm is not None else (8 << 20)
            self.g_params.make_buckets(self.bucket_bytes // 4, reverse=True)
            self.d_params.make_buckets(self.bucket_bytes // 4, reverse=False)
            if (self._overlap_in_graph() or self._comm is None) and not config.flag("GS_FORK_DIST"):
                # the pipelined four-graph iteration (see __init__); and torch.distributed's collectives, which run on THEIR stream beside the
                # launches that follow them: with the passes apart and the early contraction the world-1 step was no longer bit-identical to
                # the non-distributed one there (5e-7 on the parameters, cause not found) -- that transport keeps round 4's schedule
                self.fork = False
        K = kernels.get()
        if hasattr(K, "register_param_buffer"):  # lets the conv kernels keep their re-laid weight operands between calls
            K.register_param_buffer(self.g_params.flat)
            K.register_param_buffer(self.d_params.flat)
            K.invalidate_weights()

    def _ensure_built(self, latents, labels):
        if self.g_params is None:
            self._build(latents, labels)

    # -------------------------------------------------------------------------- inputs
    def _real_batch(self):
        """real_input_fn() -> (waveforms [B,L] | images [B,2,T,F], labels [B,61])."""
        data, labels = self.real_input_fn()
        if data.dim() == 2:  # waveforms: models.py:27-28
            images = spectral_ops.convert_to_images(data, **self.spectral_params, dtype=self.dtype)
        else:
            images = data
        return images.to(self.dtype), labels.to(self.dtype)

    # --------------------------------------------------------------------------- losses
    @staticmethod
    def _label_logits(logits, labels):
        """tf.gather_nd(logits, tf.where(labels)) for one-hot labels (models.py:39-40)."""
        # (labels are one-hot: the product is exact in the activation dtype and the fp32 sum has one non-zero term)
        return (logits * labels.to(logits.dtype)).sum(dim=1, dtype=torch.float32)

    # Each run splits into a part that touches only the network being updated and a part that needs the other network:
    #   D run:  A = D(real) + the R1 first-order pass          B = G(z) (no grad), D(fake), loss, backward
    #   G run:  A = G(z) + the mode-seeking first-order pass    B = D(G(z)), loss, backward
    # (`_a` functions return what `_b` needs).  The pipelined step overlaps the optimizer update of one network (all-reduce,
    # Adam, operand refresh on a side stream) with part A of the other network's run.
    # `fused`: the loss algebra on the [N] / [N, C] tensors -- label-logit select, softplus, penalty, mean, and their autograd
    # mirror images, ~50 torch launches per iteration -- as ONE kernel per loss (gs_gan_d_loss / gs_gan_g_loss: value and
    # gradients); `_b` then returns the mean loss itself instead of the per-sample losses.  The public *_losses methods keep
    # the per-sample form of the reference.
    def _fused_losses(self):
        # (the one-launch discriminator loss carries ONE penalty term: with the optional penalty on the generator distribution,
        # models.py:50-54 -- weight 0 in the shipped configuration -- the per-sample algebra runs instead)
        return _FUSED_LOSSES and hasattr(kernels.get(), "gan_d_loss") and not self.hyper_params.get("fake_gradient_penalty_weight", 0.0)

    def _batched_tail(self, fused, images):
        """The discriminator run sends the real and the fake batch through the latency-bound tail of the network (8x64 and below: a few
        tens of blocks per launch on 256 CUs) as ONE batch of 2n -- half the launches there, forward and backward; the R1 pass seeds the
        real rows only.  Needs the network in two pieces (networks.PGGAN.discriminator_trunk / _tail) and the one-launch loss.  (An
        activation tap still sees two passes: functional.tap_pair.)"""
        owner = getattr(self.discriminator, "__self__", None)
        # With forked branches the two passes stay apart instead: the whole fake pass (G(z), D's trunk AND tail, and through autograd their
        # backward) runs on the branch beside the real pass with its R1 passes -- twice the few-block launches of the tail, on two streams
        # that fill each other's gaps: 5.74 -> 5.42 ms against the batched tail (same box).  `batch_d_tail` overrides (tests).
        want = self.batch_d_tail if self.batch_d_tail is not None else (_BATCH_D_TAIL and not self.fork)
        return (want and fused and images.is_cuda and hasattr(owner, "discriminator_trunk")
                and getattr(self.discriminator, "__func__", None) is getattr(type(owner), "discriminator", None)
                and hasattr(kernels.get(), "lib"))

    def _d_losses_a(self, labels, real_images, fused=False):
        hp = self.hyper_params
        self._fork_mark("d_root")   # (the generator's no-grad forward of part B needs nothing of this part: it branches off here)
        real_images = real_images.detach().requires_grad_(True)
        if self._batched_tail(fused, real_images):   # part A is the real batch's trunk; everything else needs the fake batch beside it
            owner = self.discriminator.__self__
            return ("trunk", real_images) + tuple(owner.discriminator_trunk(real_images, labels.shape[1])) + (F.tap_index(),)
        _, raw = self.discriminator(real_images, labels)
        real_logits = None if fused else self._label_logits(raw, labels)
        penalty = None
        if hp.real_gradient_penalty_weight:
            with F.data_grads_only():   # tf.gradients(real_logits, [real_images]) (models.py:47): no parameter gradients on this pass
                if fused:   # d sum_i real_logit_i / d logits = the one-hot labels themselves
                    (real_gradients,) = torch.autograd.grad(raw, real_images, grad_outputs=labels.to(raw.dtype), create_graph=True)
                else:
                    (real_gradients,) = torch.autograd.grad(real_logits.sum(), real_images, create_graph=True)
            penalty = F.sumsq_rows(real_gradients)
            if not fused:   # (the fused loss kernel takes the weight itself: no scaling launch, forward or backward)
                penalty = penalty * hp.real_gradient_penalty_weight
        return (raw, penalty) if fused else (TF.softplus(-real_logits), penalty)

    def _d_losses_b(self, part_a, latents, labels, fused=False):
        hp = self.hyper_params
        if isinstance(part_a[0], str):   # ("trunk", ...): the batched-tail form of part A
            return self._d_losses_b_batched(part_a, latents, labels)
        real_part, penalty = part_a
        fake_weight = hp.get("fake_gradient_penalty_weight", 0.0)
        if fused and self._sub_runs_ok():
            return self._d_sub_runs(real_part, penalty, latents, labels)
        early, self._fake_logits_early = self._fake_logits_early, None
        if early is not None and fused:   # (the fake pass was issued in front of the real one, _issue_fake_pass_first: only the join is left)
            torch.cuda.current_stream().wait_stream(self._side)
            return F.gan_d_loss(real_part, early, labels, penalty, hp.real_gradient_penalty_weight or 1.0)
        with self._branch("d_root"):   # the whole fake pass beside the real one (its backward then runs on the branch as well)
            self._run_before_fake()
            with torch.no_grad():  # var_list is the discriminator's only: no generator backward (models.py:86-89)
                fake_images = self.generator(latents, labels)
            if fake_weight:   # tf.gradients(fake_logits, [fake_images]) (models.py:51): the images are the point of differentiation
                fake_images = fake_images.detach().requires_grad_(True)
            _, fake_logits = self.discriminator(fake_images, labels)
        if fused:
            return F.gan_d_loss(real_part, fake_logits, labels, penalty, hp.real_gradient_penalty_weight or 1.0)
        fake_logits = self._label_logits(fake_logits, labels)
        losses = real_part + TF.softplus(fake_logits)
        if penalty is not None:
            losses = losses + penalty
        if fake_weight:   # zero-centred gradient penalty on the generator distribution (models.py:50-54): the R1 kernels on the fake batch
            with F.data_grads_only():
                (fake_gradients,) = torch.autograd.grad(fake_logits.sum(), fake_images, create_graph=True)
            losses = losses + F.sumsq_rows(fake_gradients) * fake_weight
        return losses

    # The discriminator run as TWO INDEPENDENT SUB-RUNS (round 6).  L_D = mean(softplus(-r) + penalty) + mean(softplus(f)): the real pass with
    # its R1 passes and the fake pass share nothing but the parameters (leaves) -- two disjoint autograd graphs, two backward calls, one sum of
    # gradients.  With ONE loss node (gs_gan_d_loss over r and f) the real side's backward -- R1 double-backward + the real pass's own, the
    # longest chain of the run -- waits for the fake pass's forward.  Here each sub-run is issued whole: the fake sub-run on the branch from the
    # run's root (generator step pending from the previous iteration -- data parallel: its all-reduce first --, G(z), D, its loss, its backward),
    # part A of the generator run on the third stream (models._capture_merged), then the real sub-run's loss and backward on the capturing stream
    # (its forward and first-order pass were issued in part A of this run), its early contraction behind part A.  The loss VALUE is the sum of
    # the two partial means (last-bit association differs from the one-launch form; the gradients are the same numbers).
    # MEASURED, and OPT-IN (GS_SUB_RUNS=1) because of it (timelines taken from INSIDE the graph, scripts/phase_timeline.py): parity and
    # bit-identity tests green; the first form (real sub-run first, the early contraction on a stream of its own) ran 5.27 -> 5.98 ms -- four chains
    # on the runtime's four hardware queues, two of them on one; this form (fake sub-run first, part A of the generator run between the sub-runs,
    # the early contraction behind part A and behind the fake sub-run's stream: three chains from the graph's root) runs 4.91 -> 5.33 ms.  The
    # discriminator phase is throughput-bound: three chains already run side by side in the one-loss form as well.
    def _sub_runs_ok(self):
        # (also WITHOUT branches -- everything on the one stream, same order: a plain and a forked schedule then issue the same launches in the
        #  same order and stay bit-identical, tests/test_model_gpu.py::test_forked_branches_change_nothing_but_the_schedule)
        if self._forking() and not self.fork_marks:
            return False   # (the fake sub-run must start at the run's root, not behind the real sub-run's backward)
        return (self.sub_runs and self._serial_run and hasattr(F, "gan_d_loss_real") and hasattr(kernels.get(), "lib")
                and not self.hyper_params.get("fake_gradient_penalty_weight", 0.0))

    def _d_sub_runs(self, real_logits, penalty, latents, labels):
        hp = self.hyper_params
        weight = hp.real_gradient_penalty_weight or 1.0
        root = self._marks.pop("d_root", None)

        def real_sub_run():
            return F.gan_d_loss_real(real_logits, labels, penalty, weight)

        def fake_sub_run():
            self._run_before_fake()
            with torch.no_grad():  # var_list is the discriminator's only: no generator backward (models.py:86-89)
                fake_images = self.generator(latents, labels)
            _, fake_logits = self.discriminator(fake_images, labels)
            return F.gan_d_loss_fake(fake_logits, labels)

        def on_branch():
            if root is not None:
                self._marks["d_root"] = root
            return self._branch("d_root", join=False)   # (joined at the end of the run, _part_b)

        # issue order: the fake sub-run WHOLE (forward, loss, backward on the branch, from the run's root), then the real sub-run's loss and
        # backward on this stream (its forward and first-order pass were issued in part A) -- the early contraction of the real backward then
        # goes behind the fake sub-run on the branch, and the graph holds three chains: this stream, the branch, part A of the generator run
        return [(on_branch, fake_sub_run), (contextlib.nullcontext, real_sub_run)]

    def _d_losses_b_batched(self, part_a, latents, labels):
        """models.py:39-54,65 with the two discriminator passes sharing their tail: logits of [real; fake] from one pass, the R1 term
        (models.py:46-49) as the gradient of the real rows' label logits -- the cotangent of the fake rows is zero --, the loss and both
        logit gradients from the one-launch kernel."""
        hp = self.hyper_params
        _, real_images, x_real, depth, fresh, real_call = part_a
        owner = self.discriminator.__self__
        with self._branch("d_root"), torch.no_grad():  # var_list is the discriminator's only: no generator backward (models.py:86-89); beside part A
            self._run_before_fake()
            fake_images = self.generator(latents, labels)
        x_fake, depth_f, fresh_f = owner.discriminator_trunk(fake_images, labels.shape[1])
        assert depth_f == depth and fresh_f == fresh
        with F.tap_pair(real_call, F.tap_index()):
            _, raw = owner.discriminator_tail(F.cat_batch(x_real, x_fake), depth, fresh, labels, sub_batches=2)
        penalty = None
        if hp.real_gradient_penalty_weight:
            n = labels.shape[0]
            seed = self._zero_padded_rows(raw)   # (rows n..2n stay zero for the life of the buffer)
            seed[:n].copy_(labels)   # d sum_i real_logit_i / d logits: the one-hot labels on the real rows
            with F.data_grads_only():   # tf.gradients(real_logits, [real_images]) (models.py:47): no parameter gradients on this pass
                (real_gradients,) = torch.autograd.grad(raw, real_images, grad_outputs=seed, create_graph=True)
            penalty = F.sumsq_rows(real_gradients)
        return F.gan_d_loss_pair(raw, labels, penalty, hp.real_gradient_penalty_weight or 1.0)

    def _run_before_fake(self):
        hook, self._before_fake = self._before_fake, None
        if hook is not None:   # (one graph per iteration: the generator's pending update runs HERE, on the fake pass's stream, see _capture_merged)
            hook()
        if self._forking():
            self._g_ready = torch.cuda.Event()
            self._g_ready.record()

    def discriminator_losses(self, latents, labels, real_images):
        return self._d_losses_b(self._d_losses_a(labels, real_images), latents, labels)

    def _g_losses_a(self, latents, labels, fused=False):
        hp = self.hyper_params
        latents = latents.detach().requires_grad_(True)
        fake_images = self.generator(latents, labels)
        self._fork_mark("g_images")   # (the discriminator's pass over these images in part B does not wait for the first-order pass below)
        mode_seeking = None
        if hp.mode_seeking_loss_weight:
            ones = self._ones_like(fake_images)  # tf.gradients(ys) sums ys
            with F.data_grads_only():   # tf.gradients(fake_images, [latents]) (models.py:60)
                (latent_gradients,) = torch.autograd.grad(fake_images, latents, grad_outputs=ones, create_graph=True)
            if fused:
                mode_seeking = F.sumsq_rows(latent_gradients)   # (the kernel forms weight / (. + 1e-6))
            else:
                mode_seeking = 1.0 / (latent_gradients.float().pow(2).sum(dim=1) + 1.0e-6)
        return fake_images, mode_seeking

    def _g_losses_b(self, part_a, labels, fused=False):
        hp = self.hyper_params
        fake_images, mode_seeking = part_a
        if fused and mode_seeking is not None and self.split_g_loss and self._serial_run_or_merged() and hasattr(F, "gan_g_loss_mode_seeking"):
            # L_G = mean(softplus(-f)) + mean(w / (s + eps)) as TWO roots of one backward call: the mode-seeking half needs nothing of the
            # discriminator, so its second-order pass -- the longest chain of this run -- starts at the run's FIRST node, beside the
            # discriminator's forward over G(z), instead of behind a loss launch that waits for that forward (in a replayed graph a node starts
            # when its dependencies are done, whatever the order it was issued in: DESIGN.md 6.6).  One launch per half; the gradients are the
            # same numbers, the loss value is the sum of the two partial means.
            l_ms = F.gan_g_loss_mode_seeking(mode_seeking, hp.mode_seeking_loss_weight, 1.0e-6)
            with self._branch("g_images", join=False):   # (joined at the end of the run, _part_b)
                _, fake_logits = self.discriminator(fake_images, labels)
                l_adv = F.gan_g_loss(fake_logits, labels, None, 0.0, 1.0e-6)
            return (l_adv, l_ms)
        # (the same split of the DISCRIMINATOR's loss -- real half and fake half as two roots of one backward call -- measured 5.15 -> 5.41 ms: the
        #  fake pass then started 0.9 ms into the graph behind a chain it does not depend on, profiles/r06_p_split_losses_ab.txt; not kept)
        with self._branch("g_images") if mode_seeking is not None else contextlib.nullcontext():
            _, fake_logits = self.discriminator(fake_images, labels)   # (its backward then runs on the branch too: beside the second-order pass)
        if fused:
            return F.gan_g_loss(fake_logits, labels, mode_seeking, hp.mode_seeking_loss_weight, 1.0e-6)
        fake_logits = self._label_logits(fake_logits, labels)
        losses = TF.softplus(-fake_logits)
        if mode_seeking is not None:
            losses = losses + mode_seeking * hp.mode_seeking_loss_weight
        return losses

    def _serial_run_or_merged(self):
        return self._serial_run or self._nodes_on_side2 or self._capturing() or not self._forking()

    def generator_losses(self, latents, labels):
        return self._g_losses_b(self._g_losses_a(latents, labels), labels)

    # ------------------------------------------------------------------------- updates
    # Data parallelism (the reference is single-GPU): the flat gradient of a network is all-reduced in BUCKETS of
    # whole tensors (<= bucket_bytes; the generator's 16.8 MB dense weight alone), in the order the backward pass completes them,
    # and the TF-Adam update runs bucket by bucket behind its all-reduce; in eager mode the first buckets are launched from inside
    # the backward's tail (the per-layer weight-gradient contraction, kernels.flush_wgrad_reductions) as soon as their last
    # gradient is written.  The 1/world averaging is folded into the Adam kernel.
    # Two transports.  (i) Default on HIP: libgansynth_hip.so's own RCCL communicator (comm.py, gs_comm_*), every collective on
    # the backward's stream.  Nothing overlaps then -- and nothing needs an event: measured on one MI355X (RCCL, world 1, graphs):
    # 7.31 ms per iteration against 7.28 without any collective, whereas two all-reduces through torch.distributed's
    # communicator stream cost 0.43-0.49 ms of cross-stream hops (7.71-7.82 ms) before a single byte moves.  One bucket per network
    # by default (fewest launches).  (ii) torch.distributed's collectives (CPU / gloo tests, GS_TORCH_COLLECTIVES=1): asynchronous
    # on the communicator's stream, bucket k+1 on the wire under the update of bucket k.
    def _launch_reduce(self, params, bucket):
        a, b = params.buckets[bucket]
        if self._comm is not None:   # same stream as the backward: ordered by the stream itself, no event hop
            return self._comm.all_reduce_(params.grad[a:b])
        return torch.distributed.all_reduce(params.grad[a:b], async_op=True)

    def _reduce(self, params):
        """Blocking form (pipelined step): every bucket reduced, in order, on the current stream's timeline."""
        if self.distributed:
            for i in range(len(params.buckets)):
                self._launch_reduce(params, i).wait()

    def _apply(self, params, lr, beta1, beta2, reduced=False):
        params.t += 1
        lr_t = lr * math.sqrt(1.0 - beta2 ** params.t) / (1.0 - beta1 ** params.t)
        K = kernels.get()
        zero = not self.keep_gradients
        if not self.distributed or reduced:
            K.adam_tf_step(params.flat, params.grad, params.m, params.v, lr_t, beta1, beta2, 1.0e-8, 1.0 / self.world, zero_grad=zero)
            params.grad_clean = zero
            return
        works = {}
        if self._inflight is not None and self._inflight[0] is params:
            works = dict(self._inflight[1])
        self._inflight = None
        for i in range(len(params.buckets)):
            if i not in works:
                works[i] = self._launch_reduce(params, i)
        for i, (a, b) in enumerate(params.buckets):
            works[i].wait()   # (stream-side wait: the host does not block)
            K.adam_tf_step(params.flat[a:b], params.grad[a:b], params.m[a:b], params.v[a:b], lr_t, beta1, beta2, 1.0e-8,
                           1.0 / self.world, refresh=False, zero_grad=zero)
        params.grad_clean = zero   # (the buckets cover the whole buffer)
        K.invalidate_weights(params.flat)
        K.refresh_weights(params.flat)

    def _part_a(self, which, *inputs):
        """Own-network part of a run (see _d_losses_a / _g_losses_a); also arms the run: requires_grad flags, zeroed gradients."""
        if which == "d":
            self.g_params.requires_grad_(False)
            self.d_params.requires_grad_(True)
            self.d_params.begin_run()
            return self._d_losses_a(*inputs, fused=self._fused_losses())        # (labels, real_images)
        self.g_params.requires_grad_(True)
        self.d_params.requires_grad_(False)
        self.g_params.begin_run()
        return self._g_losses_a(*inputs, fused=self._fused_losses())            # (latents, labels)

    def _part_b(self, which, part_a, *inputs):
        """The rest of the run: losses, backward into the flat gradient buffer; returns the (detached) mean loss."""
        fused = self._fused_losses()
        self._branched = False
        self._origin = torch.cuda.current_stream() if torch.cuda.is_available() else None   # (the stream this run is issued -- or captured -- on)
        losses = self._d_losses_b(part_a, *inputs, fused=fused) if which == "d" else self._g_losses_b(part_a, *inputs, fused=fused)   # (latents, labels) | (labels,)
        sub_runs = losses if isinstance(losses, list) else None   # [(context factory, forward() -> root)]: see _d_sub_runs
        multi = losses if isinstance(losses, tuple) else None     # several roots of ONE backward call: see _g_losses_b
        loss = None if (sub_runs is not None or multi is not None) else (losses if losses.dim() == 0 else losses.mean())   # (the fused loss kernels return the mean itself)
        hook, self._after_loss = self._after_loss, None
        if hook is not None and sub_runs is None:
            # Merged iteration: part A of the other run forks off HERE and is issued here, in front of this run's backward (see _capture_merged).
            # (Measured round 6, profiles/r06_d_hook_after_backward_ab.txt: issued BEHIND the backward from an event recorded here: 5.23 -> 5.29 ms.)
            hook(None)
            hook = None
        self._g_ready = None
        K = kernels.get()
        deferring = _DEFER_REDUCTIONS and hasattr(K, "defer_wgrad_reductions")   # parameter gradients are only read after the whole backward:
        if deferring:                                        # their ~70 slice reductions are folded in one go at the end
            if hasattr(K, "complete_rule"):
                K.defer_wgrad_reductions(tag=which)
            else:
                K.defer_wgrad_reductions()
        params = self.d_params if which == "d" else self.g_params
        overlap = (self.distributed and deferring and len(params.buckets) > 1 and not self._capturing()
                   and not getattr(self, "_warming_up", False))
        if deferring and self.early_flush and hasattr(K, "early_flush_rule") and (self.fork or self.early_flush_always):
            # (eager launches follow the same rule, in place: a captured run and an eager one then associate every sum alike -- also the
            #  data-parallel eager path, whose buckets go on the wire from the flush at the end of the pass, behind every early launch)
            big = self._large_layer_pixels()
            if big is not None:
                self._early_in_run = 0
                K.early_flush_rule(big, self._early_flush)
        if deferring and which == "d":
            self._arm_first_bucket(K, params)
        launched = []
        if hasattr(F, "reset_fusion_state"):
            F.reset_fusion_state()   # (side-channel state of cross-node fusions is per backward pass)
        def backward(root):
            with (F.params_only() if hasattr(F, "params_only") else contextlib.nullcontext()):   # tf.gradients(loss, var_list): leaf activations want no gradient
                many = list(root) if isinstance(root, (tuple, list)) else None
                first = many[0] if many is not None else root
                if first.is_cuda and first.dim() == 0 and first.dtype == torch.float32 and not self._capturing_fresh_seed(first.device):
                    seed = F.unit_seed(first.device)   # (the loss heads recognise the seed: functional.unit_seed)
                    torch.autograd.backward(many if many is not None else root, grad_tensors=[seed] * len(many) if many is not None else seed)
                elif many is not None:
                    torch.autograd.backward(many)
                else:
                    root.backward()

        roots = []
        self._in_sub_runs = sub_runs is not None
        try:
            if multi is not None:
                backward(multi)
                roots = [r.detach() for r in multi]
            elif sub_runs is None:
                backward(loss)
            else:
                for context, forward in sub_runs:   # each on its own stream, whole: forward (what is left of it), loss, backward
                    with context():
                        root = forward()
                        if hasattr(F, "reset_fusion_state"):
                            F.reset_fusion_state()
                        backward(root)
                        roots.append(root.detach())
                    if hook is not None and self._g_ready is not None:
                        # part A of the other run: issued behind the FAKE sub-run (it starts where the generator's weights are final) and in
                        # front of the real sub-run's backward, whose early contraction queues behind it on the third stream
                        hook(self._g_ready)
                        hook = None
        finally:
            self._in_sub_runs = False
            if hasattr(F, "reset_fusion_state"):
                F.reset_fusion_state()   # (the hand-off table holds tensors of this pass -- of a graph's pool while capturing: not beyond it)
            if deferring:
                if hasattr(K, "early_flush_rule"):
                    K.early_flush_rule(0, None)
                # The final contraction reads the (x, gy) pairs and bias partial rows the branches produced and ADDS into gradients the early
                # contraction on the branch added into (read-modify-write folds, not atomics): it must sit behind the branches in the
                # captured graph, not merely behind them in time.  record_stream keeps the allocator honest and orders nothing; autograd's
                # end-of-backward sync only covers the streams of the leaves it accumulated into.  The join costs nothing: the flush needs
                # those results anyway.  (advisor, round 5)
                self._join_branches()
                if overlap:   # contract the layers bucket by bucket; a finished bucket goes on the wire under the next one's kernels
                    K.flush_wgrad_reductions(group_of=params.bucket_of,
                                             on_group_done=lambda i: launched.append((i, self._launch_reduce(params, i))))
                elif self.split_final_flush and self._forking() and self._side is not None and self._capturing():
                    K.flush_wgrad_reductions(split_stream=self._side)   # (the branch is idle here: joined above)
                else:
                    K.flush_wgrad_reductions()
        if launched:
            self._inflight = (params, launched)
        if hook is not None:   # (sub-runs: part A of the other run is issued last and starts where the generator's weights are final)
            hook(self._g_ready)
        self._join_branches()   # (a branch opened by the flush itself; a branch left open would fail the capture)
        if sub_runs is not None or multi is not None:
            loss = roots[0]
            for r in roots[1:]:
                if r.is_cuda:
                    r.record_stream(torch.cuda.current_stream())   # (made on its sub-run's stream, read here)
                loss = loss + r
        if self.distributed and self._comm is not None and self._graph_allreduce and self._capturing() and not getattr(self, "_pipe_capture", False):
            # Same-stream RCCL is capturable: the all-reduce of this run's flat gradient becomes the LAST NODE of the run's hipGraph, so
            # a replayed run hands over reduced gradients and no eager collective launch sits between the replay and the update.
            split, self._split_at = self._split_at, None
            if self._first_bucket_stream is not None:
                torch.cuda.current_stream().wait_stream(self._first_bucket_stream)
                self._first_bucket_stream = None
            if split:   # (the middle of the buffer went out behind the grouped contractions, _first_bucket_behind_groups: its two ends now)
                for a, b in ((0, split[0]), (split[1], params.grad.numel())):
                    if b > a:
                        self._comm.all_reduce_(params.grad[a:b], marker_share=(b - a) / params.grad.numel())
            else:
                self._reduce_in_capture(params)
            self._captured_reduce = True
        return loss.detach()

    def _arm_first_bucket(self, K, params):
        """Data parallel, captured discriminator run, OPT-IN (`bucket_d_reduce`, GS_DP_BUCKET_D=1): the all-reduce of the gradient in two steps.
        (Opt-in because of what it measured, DESIGN.md 7: with 300-us stand-ins for the collectives -0.09 ... -0.14 ms fully grown, +0.15 ms in a
        fade-in regime, +0.03 with 150-us ones.)
        The layers with >= 128 input channels (and the one-channel slice of the last block's conv) hold ~90 % of the bytes and sit at the BOTTOM of
        the pyramid: every pass of the backward is done with them long before it ends.  kernels.complete_rule tells when the last of their pairs
        is recorded; their contraction then runs on the branch (as the early contraction of the large layers does), and behind it, on the branch
        as well, the all-reduce of the largest range of the flat buffer that holds none of the OTHER layers' gradients -- beside the rest of the
        backward and the final contraction.  What is left on either side of that range follows where the one message went."""
        self._split_at, self._first_bucket_stream = None, None
        if not (self.bucket_d_reduce and self.distributed and self._comm is not None and self._graph_allreduce and self._capturing()
                and not self._pipe_capture and hasattr(K, "complete_rule")):
            return
        pred = lambda key: int(key[5][0]) >= 128 or int(key[5][0]) == 1   # (key: kernels._defer_wgrad; [5] = the conv input's (channels, h, w))
        named = list(params.named.items())
        sibling = {}   # weight gradient -> its bias gradient (a layer's bias follows its weight; it is complete when the layer is)
        for (name, p), (name2, p2) in zip(named, named[1:]):
            if name.endswith("/weight") and name2 == name[:-len("weight")] + "bias":
                sibling[p.grad.data_ptr()] = p2.grad
        base, size, total = params.grad.data_ptr(), params.grad.element_size(), params.grad.numel()

        def on_complete(select, others):
            def then():
                K.flush_bias_folds()
                spans = []
                for out, bias in others:
                    for t in (out, bias, sibling.get(out.data_ptr())):
                        if t is not None:
                            a, b = K._span(t)
                            spans.append(((a - base) // size, (b - base + size - 1) // size))
                if any(not (0 <= a < b <= total) for a, b in spans):
                    return   # (a gradient outside the flat buffer: the one message at the end)
                edges, at = [], 0
                for a, b in sorted(spans):
                    edges.append((at, max(at, a)))
                    at = max(at, b)
                edges.append((at, total))
                first = max(edges, key=lambda e: e[1] - e[0])
                if first[1] - first[0] > 0:
                    self._split_at = self.first_bucket = first
                    if config.flag("GS_DEBUG_DP_BUCKET"):
                        print("first bucket", first, "of", total, "behind", sum(1 for _ in others), "other layers", flush=True)
                    # on a stream of its own, behind the contraction just issued: the branch goes on to the final contraction's thin layers, and
                    # nothing of this run waits for the message before the messages at the end do (_part_b)
                    done = torch.cuda.Event()
                    done.record()
                    cur = torch.cuda.current_stream()
                    third = self._second_stream("_side3", [cur, self._origin, self._side, self._side2])
                    third.wait_event(done)
                    with torch.cuda.stream(third):
                        self._comm.all_reduce_(params.grad[first[0]:first[1]], marker_share=(first[1] - first[0]) / total)
                    self._first_bucket_stream = third
            self._early_flush(select, then=then)
        K.complete_rule(pred, on_complete)

    def _join_branches(self):
        """The current stream waits for every branch this run opened: the side stream (every branch was joined where it closed, except the
        early contraction's, and autograd joins the streams it used) and, in the merged iteration, the stream this run's own-network nodes
        ran -- and accumulated -- on."""
        if self._branched:
            self._branched = False
            torch.cuda.current_stream().wait_stream(self._side)
        if (self._nodes_on_side2 or self._early_on_side2) and self._forking():
            self._early_on_side2 = False
            torch.cuda.current_stream().wait_stream(self._side2)

    def _reduce_in_capture(self, params):
        """The gradient all-reduce issued while the current stream is being captured into a hipGraph (a method of its own so that a
        test can make it raise and watch every rank fall back together)."""
        self._reduce(params)

    def _agree(self, ok):
        """Data parallel: did EVERY rank succeed?  A rank-local failure (allocator, capture) must not leave one rank on a different
        launch sequence than its peers -- their collectives would no longer pair up and the job would hang -- so the outcome of
        anything that may fail locally is agreed on with an eager MIN all-reduce over the launcher's process group, outside any
        capture, and every rank takes the same branch."""
        if not self.distributed or self.world <= 1:
            return bool(ok)
        dev = self.g_params.flat.device
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item())

    def _give_up_graph_collectives(self, which, error):
        """Every rank lands here together (see _agree): no collective inside captured graphs any more.  A capture that aborted with an
        ncclAllReduce inside may have left our communicator unusable, so the eager collectives move to torch.distributed's own."""
        import sys
        print("gansynth_amd.models: capturing the gradient all-reduce inside the %s run's graph failed on some rank (here: %s); "
              "it will run eagerly after each replay" % (which, "ok" if error is None else str(error).splitlines()[0]), file=sys.stderr, flush=True)
        self._graph_allreduce = False
        self._captured_reduce = False
        # Every graph captured so far may replay an ncclAllReduce on the communicator given up here (the OTHER run's graph of the
        # serial path, the pairs of the pipelined step): all of them go, so that every run is captured again without a collective.
        torch.cuda.synchronize()
        self._graphs.clear()
        self._merged = None
        if self._pipe is not None:
            self._pipe.pop("d", None), self._pipe.pop("g", None)
            self._pipe["key"] = None
        if self.world > 1 and self._comm is not None:
            # Not destroyed: ncclCommDestroy on a communicator an aborted capture left half-enqueued may block.  It is retired --
            # never used again, kept alive until the process ends -- and the eager collectives go through torch.distributed.
            self._retired_comm = self._comm
            self._comm = None
        self._abandon_capture(which)

    def _abandon_capture(self, which):
        """State left behind by a _forward_backward that raised in the middle of a stream capture: deferred kernel-layer jobs, half-built
        fusion hand-offs and the gradients the partial backward wrote."""
        torch.cuda.synchronize()
        K = kernels.get()
        if hasattr(K, "drop_deferred"):
            K.drop_deferred()
        F.reset_fusion_state()
        self._inflight = None
        self._after_loss = None
        self._marks.clear()
        params = self.d_params if which == "d" else self.g_params
        if not self.keep_gradients:   # (as before the first capture: a graph without a fill must find the buffer the way every replay will)
            params.grad.zero_()
            params.grad_clean = True

    @staticmethod
    def _capturing():
        return torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()

    def _ones_like(self, t):
        """A constant all-ones tensor of t's shape, layout and dtype, filled once (never written afterwards; a fill created inside a
        stream capture would belong to that graph's pool, so there the plain ones_like runs)."""
        cache = self.__dict__.setdefault("_ones_cache", {})
        key = (tuple(t.shape), tuple(t.stride()), t.dtype, str(t.device))
        ones = cache.get(key)
        if ones is None:
            ones = torch.ones_like(t)
            if not self._capturing():
                cache[key] = ones
        return ones

    def _zero_padded_rows(self, t):
        """A tensor of t's shape whose rows the caller overwrites only in the upper half; zeroed once (same rule as _ones_like)."""
        cache = self.__dict__.setdefault("_zero_cache", {})
        key = (tuple(t.shape), t.dtype, str(t.device))
        z = cache.get(key)
        if z is None:
            z = torch.zeros_like(t)
            if not self._capturing():
                cache[key] = z
        return z

    def _capturing_fresh_seed(self, device):
        """True when the constant seed of this device would have to be CREATED inside a stream capture (its memory would belong
        to that graph's pool): then the plain loss.backward() runs."""
        if str(torch.device(device)) in F._UNIT:
            return False
        if self._capturing():
            return True
        F.unit_seed(device)
        return False

    def _forward_backward(self, which, *inputs):
        """Gradients of one run into the flat gradient buffer; returns the (detached) mean loss.
        inputs: (latents, labels, real_images) for "d", (latents, labels) for "g"."""
        self._serial_run = True   # (both parts inside one capture: a branch of part B may start at a mark of part A)
        try:
            if which == "d":
                latents, labels, real_images = inputs
                if self._d_fake_first():
                    self._issue_fake_pass_first(latents, labels)
                return self._part_b("d", self._part_a("d", labels, real_images), latents, labels)
            latents, labels = inputs
            return self._part_b("g", self._part_a("g", latents, labels), labels)
        finally:
            self._serial_run = False
            self._marks.clear()

    # One graph per iteration, opt-in (GS_FAKE_FIRST=1 / `fake_first`): the discriminator run's FAKE pass is issued before the real pass.  It
    # starts with the generator's pending step, i.e. data parallel with the all-reduce of the generator's gradient; issued first it is the
    # capturing stream's own chain and the real pass the branch.  One GPU: 5.17 -> 5.21 ms.  World size 1 with 300-us stand-ins for the two
    # collectives (scripts/dp_marker_check.py, profiles/r06_g_dp_markers.txt): the stand-ins add 0.43 ms with this order against 0.56 real-first
    # and 0.61 in the two-graph form in one process, 0.61 against 0.52 inside tests/test_model_gpu.py's process -- the discriminator phase is
    # throughput-bound and which chains share a hardware queue depends on the stream pool's history: not a gain to build a
    # default on.
    def _d_fake_first(self):
        if not (self._forking() and self._capturing() and self.fork_marks and self._fused_losses()) or self._sub_runs_ok():
            return False
        if self.batch_d_tail:   # (tests: real and fake through the tail as one batch -- there is no separate fake pass then)
            return False
        want = self.fake_first
        return bool(want) and not self.hyper_params.get("fake_gradient_penalty_weight", 0.0)

    def _issue_fake_pass_first(self, latents, labels):
        self.g_params.requires_grad_(False)   # (what _part_a("d") arms: the discriminator's nodes must be recorded for its backward)
        self.d_params.requires_grad_(True)
        self._fork_mark("d_root")
        with self._branch("d_root", join=False):   # (joined where the loss needs it, _d_losses_b)
            self._run_before_fake()
            with torch.no_grad():  # var_list is the discriminator's only: no generator backward (models.py:86-89)
                fake_images = self.generator(latents, labels)
            _, fake_logits = self.discriminator(fake_images, labels)
        self._fake_logits_early = fake_logits

    def _regime(self):
        """(head depth, fade weight or None) of the networks at the current growing depth; None for foreign network objects."""
        owner = getattr(self.generator, "__self__", None)
        if owner is None or not hasattr(owner, "_head_depth") or getattr(self.discriminator, "__self__", None) is not owner:
            return None
        return owner._head_depth(owner.growing_depth)

    def _graphable(self):
        """hipGraph replay needs a step-invariant launch sequence: the network structure is fixed within a growing regime (head
        depth, faded or not -- graphs are re-captured when it changes) and the one per-step scalar, the fade-in weight, is read
        from device memory (functional.DeviceLerp)."""
        return self.use_graphs and torch.cuda.is_available() and self._regime() is not None

    def _fully_grown(self):
        reg = self._regime()
        owner = getattr(self.generator, "__self__", None)
        return reg is not None and reg[1] is None and reg[0] == owner.max_depth

    def _run(self, which, *inputs):
        self._join_updates()
        owner = getattr(self.generator, "__self__", None)
        self._run_reduced = False
        if not self._graphable():
            self._graphs.clear()
            with self._stream_guard() if self.fork_eager else contextlib.nullcontext():
                return self._forward_backward(which, *inputs)
        head, fade = self._regime()
        key = (head, fade is None)
        if self._graph_key != key:   # a new growing regime: different launch sequence
            self._graphs.clear()
            self._graph_key = key
            F.drop_constants()   # (junction constants of the old regime's shapes; live graphs hold their own references)
        if fade is not None:
            if self._lerp is None:
                self._lerp = F.DeviceLerp(self.g_params.flat.device)
            self._lerp.set(fade)   # (stream-ordered before the replay below)
        entry = self._graphs.get(which)
        if entry is not None and entry[4] != self.keep_gradients:
            entry = None   # (a graph captured without a gradient fill relies on the zeroing optimizer step behind every replay)
        if entry is None or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(entry[1], inputs)):
            self._check_fork_runtime()
            static = [t.detach().clone() for t in inputs]
            K = kernels.get()
            owner.fade_weight = self._lerp if fade is not None else None   # the networks read the weight from the device table
            try:
                side = torch.cuda.Stream()
                side.wait_stream(torch.cuda.current_stream())
                self._warming_up = True
                try:
                    with torch.cuda.stream(side):  # one eager pass on a side stream (allocator / lazy-init warm-up)
                        self._forward_backward(which, *static)
                        if self.distributed and self._comm is not None and self._graph_allreduce:
                            # RCCL sets up its channels on the first collective of a kind: not capturable, so one eager all-reduce
                            # of the buffers the graph will reduce (every rank does the same; the gradients are dead values here)
                            self._reduce(self.d_params if which == "d" else self.g_params)
                finally:
                    self._warming_up = False
                torch.cuda.current_stream().wait_stream(side)
                # the warm-up pass left its gradients in the flat buffer and no optimizer step clears them: a graph that relies on the
                # step's clearing (keep_gradients = False: no fill inside) must find the buffer as every later replay will
                params_ = self.d_params if which == "d" else self.g_params
                if not self.keep_gradients:
                    params_.grad.zero_()
                    params_.grad_clean = True
                # the prepared weight operands live in persistent workspaces that the optimizer step refreshes eagerly
                # (kernels.adam_tf_step): bring them up to date now so that the captured graph holds no re-layout launches
                K.refresh_weights()
                graph = torch.cuda.CUDAGraph()
                self._captured_reduce = False
                with_collective = self.distributed and self._comm is not None and self._graph_allreduce
                error = None
                try:
                    with _quiet_gc(), self._leveled_queues(), self._stream_guard(), torch.cuda.graph(graph, **_capture_mode(with_collective, self.fork)):
                        loss = self._forward_backward(which, *static)
                except RuntimeError as e:
                    if not with_collective:
                        raise
                    error = e
                if with_collective and not self._agree(error is None):
                    # The collective would not go into the graph on SOME rank: every rank (agreed above, so that no rank keeps a graph
                    # with the collective inside while a peer reduces eagerly) captures the run again without it -- the all-reduce
                    # then follows each replay eagerly, as in round 2.
                    self._give_up_graph_collectives(which, error)
                    graph = torch.cuda.CUDAGraph()
                    with _quiet_gc(), self._leveled_queues(), self._stream_guard(), torch.cuda.graph(graph):
                        loss = self._forward_backward(which, *static)
            finally:
                owner.fade_weight = None   # (only captured launches use the table; eager callers keep passing the number)
            entry = (graph, static, loss, self._captured_reduce, self.keep_gradients, F.constants_snapshot())
            self._graphs[which] = entry
        graph, static, loss, reduced = entry[:4]
        if not self.keep_gradients:
            params_ = self.d_params if which == "d" else self.g_params
            if not params_.grad_clean:   # (a replay not preceded by the zeroing update: e.g. a run repeated without its optimizer step)
                params_.grad.zero_()
                params_.grad_clean = True
            params_.grad_clean = False   # (what begin_run did at capture time: the replay accumulates into the buffer)
        _copy_inputs(static, inputs)
        graph.replay()
        self._run_reduced = reduced   # (the replay already summed the gradients over the ranks: _apply goes straight to the update)
        return loss

    def discriminator_step(self, latents, labels, real_images):
        self._ensure_built(latents, labels)
        hp = self.hyper_params
        loss = self._run("d", latents, labels, real_images)
        self._apply(self.d_params, hp.discriminator_learning_rate, hp.discriminator_beta1, hp.discriminator_beta2, reduced=self._run_reduced)
        self.discriminator_loss = loss
        return self.discriminator_loss

    def generator_step(self, latents, labels):
        self._ensure_built(latents, labels)
        hp = self.hyper_params
        loss = self._run("g", latents, labels)
        self._apply(self.g_params, hp.generator_learning_rate, hp.generator_beta1, hp.generator_beta2, reduced=self._run_reduced)
        self.global_step += 1  # models.py:84
        self.generator_loss = loss
        return self.generator_loss

    # ------------------------------------------------------------------ pipelined iteration
    # Every run as TWO graphs: part A (own network only) and part B (the rest), so that the optimizer update of the OTHER network --
    # its gradient all-reduce above all -- can sit between them:
    #     D.A | update G | D.B | G.A | update D | G.B
    # Part A needs neither the gradients being reduced nor the parameters about to change.  Two ways to overlap the collective
    # with part A:
    #   (i) IN THE GRAPH (data parallel on our own RCCL communicator, the default there): the all-reduce of the other network's flat
    #       gradient is a forked branch INSIDE graph A -- fork at the graph's root, join at its end -- so the collective node is off
    #       the critical path of part A's kernels and there is no cross-stream event between replays (an event hop between a replay
    #       and another stream costs 0.25-0.75 ms on this stack, scripts/cross_stream_cost.py).  Adam and the operand refresh stay
    #       eager on the main stream behind graph A (lr_t is a by-value scalar; streaming kernels beside the persistent conv blocks
    #       cost the main stream 5 %, measured).
    #   (ii) SIDE STREAM (opt-in, GS_PIPELINE=1 with torch.distributed's collectives): the round-2 form.
    def _join_updates(self):
        """A pipelined step leaves the generator's update pending: apply it (reducing the gradient first when the all-reduce was
        going to ride in the next discriminator graph).  Data parallel: with a reduction pending this IS a collective -- every rank
        must get here at the same point of its launch sequence.  train() therefore joins on EVERY rank before a rank-0 checkpoint
        and at its end; synchronize(), state_dict / checkpoint.save and generate() called by hand on a distributed model must be
        called on all ranks (`collective_pending()` tells whether the call would communicate)."""
        if self._g_pending is not None:   # (one graph per iteration: the generator's update rides at the front of the NEXT graph -- or here)
            lr_t, self._g_pending = self._g_pending, None
            hp = self.hyper_params
            if self.distributed:
                self._reduce(self.g_params)
            zero = not self.keep_gradients
            kernels.get().adam_tf_step(self.g_params.flat, self.g_params.grad, self.g_params.m, self.g_params.v, lr_t, hp.generator_beta1,
                                       hp.generator_beta2, 1.0e-8, 1.0 / self.world, zero_grad=zero)
            self.g_params.grad_clean = zero
        if self._pipe is not None and self._pipe.get("g_pending"):
            hp = self.hyper_params
            P = self._pipe
            P["g_pending"] = False
            if P.get("g_unreduced"):
                P["g_unreduced"] = False
                self._reduce(self.g_params)
            else:
                torch.cuda.current_stream().wait_event(P["g_reduced"])
            self._apply(self.g_params, hp.generator_learning_rate, hp.generator_beta1, hp.generator_beta2, reduced=True)

    def collective_pending(self):
        """True when the next _join_updates() / synchronize() / generate() / checkpoint would issue a gradient all-reduce (the
        pipelined data-parallel step leaves the generator's gradient unreduced until the next discriminator graph)."""
        if self.distributed and self.world > 1 and self._g_pending is not None:
            return True
        return bool(self.distributed and self.world > 1 and self._pipe is not None and self._pipe.get("g_pending")
                    and self._pipe.get("g_unreduced"))

    def synchronize(self):
        """Everything a train_step enqueued (including the pending update) has finished.  Collective when `collective_pending()`."""
        self._join_updates()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    def _overlap_in_graph(self):
        """The gradient all-reduce as a forked branch of the other run's part-A graph: data parallel, own RCCL communicator, in-graph
        collectives not refused (GS_NO_GRAPH_ALLREDUCE / a failed capture), not switched off (GS_NO_OVERLAP_REDUCE=1)."""
        return self.distributed and self._comm is not None and self._graph_allreduce and self.overlap_reduce

    def _capture_pair(self, which, a_inputs, b_inputs, reduce_params=None):
        """Two graphs for one run: part A (own network) and part B (the rest), sharing one memory pool (replayed A, B, A, B ...).
        `reduce_params`: the OTHER network's parameters, whose flat gradient is all-reduced on a forked branch of graph A."""
        K = kernels.get()
        owner = getattr(self.generator, "__self__", None)
        _, fade = self._regime()
        sa = [t.detach().clone() for t in a_inputs]
        sb = [t.detach().clone() for t in b_inputs]
        params = self.d_params if which == "d" else self.g_params
        owner.fade_weight = self._lerp if fade is not None else None   # the networks read the fade weight from the device table
        self._pipe_capture = True   # (the pipelined step places its reductions itself: none at the end of part B)
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            self._warming_up = True
            try:
                with torch.cuda.stream(side):  # one eager pass on a side stream (allocator / lazy-init warm-up)
                    self._part_b(which, self._part_a(which, *sa), *sb)
                    if reduce_params is not None:   # RCCL sets up its channels on the first collective: not capturable (dead values here)
                        self._reduce(reduce_params)
            finally:
                self._warming_up = False
            torch.cuda.current_stream().wait_stream(side)
            if not self.keep_gradients:   # the graphs hold no fill: they rely on the zeroing optimizer step behind every part B (see _run)
                params.grad.zero_()
                params.grad_clean = True
            K.refresh_weights()   # (see _run: the captured graphs hold no re-layout launches)
            ga = torch.cuda.CUDAGraph()
            import warnings
            with warnings.catch_warnings(record=True) as caught:
                warnings.simplefilter("always")
                with _quiet_gc(), self._leveled_queues(), self._stream_guard(), torch.cuda.graph(ga, **_capture_mode(reduce_params is not None, self.fork)):
                    if reduce_params is not None:
                        main = torch.cuda.current_stream()
                        fork = torch.cuda.Stream()
                        while fork.cuda_stream == main.cuda_stream or (self._side is not None and fork.cuda_stream == self._side.cuda_stream):
                            fork = torch.cuda.Stream()   # (pooled streams come round-robin: never the capturing one, nor the branches')
                        fork.wait_stream(main)            # fork at the root of the graph ...
                        with torch.cuda.stream(fork):
                            self._reduce_in_capture(reduce_params)
                    part_a = self._part_a(which, *sa)
                    if reduce_params is not None:
                        main.wait_stream(fork)            # ... join at its end: the collective runs beside all of part A
            # Part A may hold NO kernel: a discriminator whose whole depth runs in the batched tail has no trunk of its own (shallow
            # growing regimes), and on ONE rank RCCL short-cuts the all-reduce to nothing as well.  torch warns about the empty graph;
            # that is the only way it can be empty -- with peers the collective is a node -- and an empty part A is simply not replayed.
            a_empty = any("Graph is empty" in str(w.message) for w in caught)
            for w in caught:
                if "Graph is empty" not in str(w.message):
                    warnings.warn_explicit(w.message, w.category, w.filename, w.lineno)
            if a_empty and reduce_params is not None and self.world > 1:
                raise RuntimeError("part A of the %s run captured no node although it holds a gradient all-reduce over %d ranks" % (which, self.world))
            gb = torch.cuda.CUDAGraph()
            with _quiet_gc(), self._leveled_queues(), self._stream_guard(), torch.cuda.graph(gb, pool=ga.pool()):
                loss = self._part_b(which, part_a, *sb)
        finally:
            self._pipe_capture = False
            owner.fade_weight = None
        return {"a": ga, "b": gb, "sa": sa, "sb": sb, "loss": loss, "reduces": reduce_params is not None, "keep": self.keep_gradients, "a_empty": a_empty,
                "consts": F.constants_snapshot()}   # (cached junction constants the graphs read: alive as long as the graphs)

    def _pipelined_ok(self):
        if not self._graphable():
            return False
        if self._overlap_in_graph():
            return True
        return self.pipeline and self._fully_grown()

    def _train_step_pipelined(self, d_latents, d_labels, real_images, g_latents, g_labels):
        """D.A | update G | D.B | G.A | update D | G.B (see above)."""
        hp = self.hyper_params
        main = torch.cuda.current_stream()
        if self._pipe is None:
            ev = lambda: torch.cuda.Event()
            self._pipe = {"side": torch.cuda.Stream(), "d_done": ev(), "g_done": ev(), "d_reduced": ev(), "g_reduced": ev(), "g_pending": False,
                          "g_unreduced": False, "key": None}
        P = self._pipe
        in_graph = self._overlap_in_graph()
        use_side = (not in_graph) and (self.pipe_side if self.pipe_side is not None else (self.distributed if _PIPE_SIDE is None else _PIPE_SIDE))
        head, fade = self._regime()
        key = (head, fade is None, in_graph, self.keep_gradients)
        if fade is not None:
            if self._lerp is None:
                self._lerp = F.DeviceLerp(self.g_params.flat.device)
            self._lerp.set(fade)   # (stream-ordered before the replays below)

        def fresh(entry, a_inputs, b_inputs):
            return entry is None or any(x.shape != y.shape or x.dtype != y.dtype for x, y in zip(entry["sa"] + entry["sb"], list(a_inputs) + list(b_inputs)))

        d_in = ((d_labels, real_images), (d_latents, d_labels))
        g_in = ((g_latents, g_labels), (g_labels,))
        if P["key"] != key or fresh(P.get("d"), *d_in) or fresh(P.get("g"), *g_in):
            self._join_updates()
            self._graphs.clear()
            P.pop("d", None), P.pop("g", None)
            if P["key"] is not None and P["key"][:2] != key[:2]:
                F.drop_constants()   # (a new growing regime: see _run)
            error = None
            try:
                P["d"] = self._capture_pair("d", *d_in, reduce_params=self.g_params if in_graph else None)
                P["g"] = self._capture_pair("g", *g_in, reduce_params=self.d_params if in_graph else None)
            except RuntimeError as e:
                if not in_graph:
                    raise
                error = e
            if in_graph and not self._agree(error is None):
                # some rank could not capture the collective: EVERY rank drops the in-graph form (agreed, so that the collective
                # sequences of the ranks stay identical) and captures plain pairs; the reductions then run eagerly between replays
                self._give_up_graph_collectives("d", error)
                self._abandon_capture("g")
                in_graph = False
                use_side = self.pipe_side if self.pipe_side is not None else False
                key = (head, fade is None, in_graph, self.keep_gradients)
                P["d"] = self._capture_pair("d", *d_in)
                P["g"] = self._capture_pair("g", *g_in)
            P["key"] = key
        D, G = P["d"], P["g"]
        _copy_inputs(D["sa"] + D["sb"] + G["sa"] + G["sb"], list(d_in[0]) + list(d_in[1]) + list(g_in[0]) + list(g_in[1]))

        def armed(params):
            """A no-fill graph is about to accumulate into this buffer: it must be clean (the zeroing update behind the last part B)."""
            if not self.keep_gradients:
                if not params.grad_clean:
                    params.grad.zero_()
                params.grad_clean = False

        if in_graph:
            # D run.  Graph A = {D part A  ||  all-reduce of the generator's pending gradient}; then the generator's update (part B runs
            # the generator), then part B.
            armed(self.d_params)
            if not D["a_empty"]:
                D["a"].replay()
            if P["g_pending"]:
                P["g_pending"] = P["g_unreduced"] = False
                self._apply(self.g_params, hp.generator_learning_rate, hp.generator_beta1, hp.generator_beta2, reduced=True)
            D["b"].replay()
            # G run.  Graph A = {G part A  ||  all-reduce of the discriminator's gradient}; the discriminator's update; part B runs it.
            armed(self.g_params)
            if not G["a_empty"]:
                G["a"].replay()
            self._apply(self.d_params, hp.discriminator_learning_rate, hp.discriminator_beta1, hp.discriminator_beta2, reduced=True)
            G["b"].replay()
            P["g_pending"] = P["g_unreduced"] = True   # reduced inside the next D graph (or eagerly by _join_updates)
            self.global_step += 1  # models.py:84
            self.discriminator_loss, self.generator_loss = D["loss"], G["loss"]
            return D["loss"], G["loss"]

        def reduce_async(params, done, reduced):
            done.record(main)
            if use_side:
                with torch.cuda.stream(P["side"]):
                    P["side"].wait_event(done)
                    self._reduce(params)
                    reduced.record(P["side"])
            else:
                self._reduce(params)
                reduced.record(main)

        # D run.  Part A reads the discriminator only: it runs under the all-reduce of the previous generator gradients.
        armed(self.d_params)
        if not D["a_empty"]:
            D["a"].replay()
        self._join_updates()                        # the generator's update; part B runs the generator
        D["b"].replay()
        reduce_async(self.d_params, P["d_done"], P["d_reduced"])
        # G run.  Part A reads the generator only: it runs under the all-reduce of the discriminator's gradients.
        armed(self.g_params)
        if not G["a_empty"]:
            G["a"].replay()
        main.wait_event(P["d_reduced"])
        self._apply(self.d_params, hp.discriminator_learning_rate, hp.discriminator_beta1, hp.discriminator_beta2, reduced=True)
        G["b"].replay()                             # runs the updated discriminator
        reduce_async(self.g_params, P["g_done"], P["g_reduced"])
        P["g_pending"] = True                       # applied before the next part B (or by _join_updates / synchronize)
        P["g_unreduced"] = False
        self.global_step += 1  # models.py:84
        self.discriminator_loss, self.generator_loss = D["loss"], G["loss"]
        return D["loss"], G["loss"]

    # ------------------------------------------------------------------- merged iteration
    # One GPU, graphs with branches: part A of the GENERATOR run (G(z) and the mode-seeking first-order pass: its own network only, whose
    # parameters the discriminator run does not touch) is captured INSIDE the discriminator run's graph, on a stream of its own from the
    # graph's root -- the discriminator run's second half is one stream wide (the fake pass and the early weight gradients are done, the R1
    # double-backward, the real pass's backward and the final contraction remain), and the generator's few-block levels fill from it and
    # into it.  Two graphs per iteration as before:   X = { D run  ||  G.A }   update D   Y = { G.B }   update G.
    # 5.13 / 5.07 -> 4.99 / 4.93 ms on one box (first measured neutral, 5.19-5.23 against 5.22-5.24: the two passes of the discriminator
    # run were still waiting on each other at every accumulate target, kernels._adds_into).
    # The generator's own-network nodes were created on that stream, so autograd runs their backward there in Y as well (joined at the
    # end of the run, _part_b).
    def _merged_ok(self):
        # (data parallel: with the gradient all-reduce as the last node of each of the two graphs -- own communicator, in-graph collectives
        #  not refused, not the four-graph overlapped form)
        dp_ok = not self.distributed or (self._comm is not None and self._graph_allreduce and not self._overlap_in_graph())
        return (self.merge_runs and self.fork and self._graphable() and dp_ok and self._fused_losses() and hasattr(kernels.get(), "lib"))

    # One graph per iteration (round 6; `fuse_iteration`, GS_NO_FUSED_ITERATION=1 returns to the pair above).  The two optimizer steps were the
    # only eager launches left between the graphs -- lr_t is a by-value scalar -- and with them outside, (i) every iteration pays two graph
    # boundaries, (ii) nothing can run beside an update, and (iii) data parallel, an all-reduce can only be the LAST node of a graph: exposed.
    # gs_adam_tf_step_dev reads lr_t from device memory, so the whole iteration is ONE graph Z:
    #     Z_k = { D real pass + R1 first-order pass        ||  [all-reduce G_{k-1}] -> Adam G_{k-1} -> refresh G -> D run's fake pass }
    #           -> D loss -> { D backward, contraction [all-reduce D_k] -> Adam D_k -> refresh D   ||  G.A_k }  ->  G.B_k
    # The GENERATOR's update of iteration k - 1 rides at the front of Z_k on the fake pass's branch: the discriminator's real pass and its R1
    # passes need nothing of the generator, and the fake pass (G fwd + D fwd + D bwd = 4 network passes against ~7 on the real side) has the
    # slack.  Data parallel this is where the generator's all-reduce hides by construction.  The discriminator's update sits where it always
    # did -- behind its backward -- but part A of the generator run is still in flight beside it.  After Z_k the generator's gradient is
    # PENDING (`_g_pending` holds its lr_t): the next replay applies it (lr slot >= 0), anything else that needs the weights -- generate(),
    # a checkpoint, a run outside this path, a new growing regime -- goes through _join_updates() first.  A freshly captured Z finds no
    # pending step: its lr slot is negative and the kernel leaves every buffer untouched.
    def _fused_ok(self):
        return self.fuse_iteration and hasattr(kernels.get(), "adam_tf_step_dev")

    @staticmethod
    def _lr_t(lr, beta1, beta2, t):
        return lr * math.sqrt(1.0 - beta2 ** t) / (1.0 - beta1 ** t)

    def _apply_in_graph(self, params, slot, beta1, beta2, reduce_first=False):
        """The optimizer step as nodes of the graph being captured: [all-reduce] -> Adam with lr_t from the device table -> operand refresh."""
        if reduce_first:
            self._reduce_in_capture(params)
        kernels.get().adam_tf_step_dev(params.flat, params.grad, params.m, params.v, self._opt_scalars.ptr(slot), beta1, beta2, 1.0e-8,
                                       1.0 / self.world, zero_grad=not self.keep_gradients)

    def _capture_merged(self, d_inputs, g_inputs, fused=False):
        K = kernels.get()
        owner = getattr(self.generator, "__self__", None)
        _, fade = self._regime()
        sd = [t.detach().clone() for t in d_inputs]
        sg = [t.detach().clone() for t in g_inputs]
        owner.fade_weight = self._lerp if fade is not None else None   # the networks read the fade weight from the device table
        with_collective = self.distributed and self._comm is not None and self._graph_allreduce
        error, reduced = None, [False, False]
        try:
            warm = torch.cuda.Stream()
            warm.wait_stream(torch.cuda.current_stream())
            self._warming_up = True
            try:
                with torch.cuda.stream(warm):  # one eager pass on a side stream (allocator / lazy-init warm-up)
                    self._forward_backward("d", *sd)
                    self._forward_backward("g", *sg)
                    if with_collective:   # RCCL sets up its channels on the first collective of a kind: not capturable (dead values here)
                        self._reduce(self.d_params)
                        self._reduce(self.g_params)
            finally:
                self._warming_up = False
            torch.cuda.current_stream().wait_stream(warm)
            if not self.keep_gradients:   # the graphs hold no fill: they rely on the zeroing optimizer steps (see _run)
                for params in (self.d_params, self.g_params):
                    params.grad.zero_()
                    params.grad_clean = True
            K.refresh_weights()   # (see _run: the captured graphs hold no re-layout launches)
            hp = self.hyper_params
            if fused:
                if self._opt_scalars is None:
                    self._opt_scalars = F.DeviceScalars(self.g_params.flat.device, 2)
                # the per-network refresh launches read descriptor tables that are built (host -> device) on first use: not inside a capture
                for params in (self.d_params, self.g_params):
                    K.invalidate_weights(params.flat)
                    K.refresh_weights(params.flat)
            try:
                gx = torch.cuda.CUDAGraph()
                self._captured_reduce = False
                with _quiet_gc(), self._leveled_queues(), self._stream_guard(), torch.cuda.graph(gx, **_capture_mode(with_collective, self.fork)):
                    main = torch.cuda.current_stream()
                    side2 = self._second_stream("_side2", [main, self._side])
                    box = []
                    if fused:
                        # the generator's PENDING step at the front of the fake pass's branch (no stream of its own: the branch is its only
                        # consumer until the join, and a graph one branch wider would need one more of the runtime's four hardware queues)
                        self._before_fake = lambda: self._apply_in_graph(self.g_params, 1, hp.generator_beta1, hp.generator_beta2,
                                                                         reduce_first=with_collective)

                    def part_a_of_g(mark=None):
                        # from the discriminator run's loss on its second half is one stream wide (R1 double-backward, the real pass's backward,
                        # the final contraction): part A of the generator run goes THERE (from the graph's root, beside the two forward passes,
                        # measured 5.27 -> 5.34 ms in round 5).  `mark`: an event recorded at the loss -- the branch starts there although it is issued later.
                        if mark is not None:
                            side2.wait_event(mark)
                        else:
                            side2.wait_stream(torch.cuda.current_stream())
                        with torch.cuda.stream(side2):
                            box.append(self._part_a("g", *sg))
                        self.g_params.requires_grad_(False)      # (back to the discriminator run's arming for its backward)
                        self.d_params.requires_grad_(True)
                    self._after_loss = part_a_of_g
                    d_loss = self._forward_backward("d", *sd)    # (data parallel: ends with the all-reduce of the discriminator's gradient, _part_b)
                    g_part_a = box[0]
                    if fused:
                        if self._before_fake is not None:
                            raise RuntimeError("the discriminator run never reached its fake pass: the generator's pending step has no place in the graph")
                        # the discriminator's step, behind its (all-reduced) gradient; part A of the generator run is still in flight beside it
                        self._apply_in_graph(self.d_params, 0, hp.discriminator_beta1, hp.discriminator_beta2)
                    main.wait_stream(side2)                      # ... join at its end
                    if fused:   # part B of the generator run in the same graph (reads the discriminator just updated)
                        reduced[0] = self._captured_reduce
                        self.g_params.requires_grad_(True)
                        self.d_params.requires_grad_(False)
                        self._side2.wait_stream(main)            # (as in the pair's second graph: the generator's nodes run on their stream again)
                        self._nodes_on_side2 = True
                        self._pipe_capture = True                # (no all-reduce at the end of THIS run: it opens the next graph, beside the real pass)
                        try:
                            g_loss = self._part_b("g", g_part_a, sg[1])
                        finally:
                            self._nodes_on_side2 = False
                            self._pipe_capture = False
                        reduced[1] = reduced[0]
                if not fused:
                    reduced[0] = self._captured_reduce
                gy = None if fused else torch.cuda.CUDAGraph()
                self._captured_reduce = False
                with (contextlib.nullcontext() if fused else contextlib.ExitStack()) as stack:
                    if not fused:
                        for cm in (_quiet_gc(), self._leveled_queues(), self._stream_guard(),
                                   torch.cuda.graph(gy, pool=gx.pool(), **_capture_mode(with_collective, self.fork))):
                            stack.enter_context(cm)
                        self.g_params.requires_grad_(True)           # (the discriminator run in between armed the other network)
                        self.d_params.requires_grad_(False)
                        # the generator's nodes will run on their stream again (autograd): it joins THIS capture here, from the root, as a child of
                        # the capturing stream -- joining later through an event of the other branch (the discriminator's gradient arrives from
                        # there) made the two branches each other's parent and hip::Stream::EndCapture recursed until the stack ran out
                        self._side2.wait_stream(torch.cuda.current_stream())
                        self._nodes_on_side2 = True
                        try:
                            g_loss = self._part_b("g", g_part_a, sg[1])
                        finally:
                            self._nodes_on_side2 = False
                if not fused:
                    reduced[1] = self._captured_reduce
            except RuntimeError as e:
                if not with_collective:
                    raise
                error = e
            if with_collective and not self._agree(error is None):
                # the collective would not go into a graph on SOME rank: every rank (agreed) drops the in-graph form; the iteration then runs as
                # two plain runs with the all-reduce eagerly behind each replay (_run)
                self._give_up_graph_collectives("d", error)
                self._abandon_capture("g")
                return None
        finally:
            owner.fade_weight = None
            self._after_loss = None   # (a capture that raised before the discriminator run's loss must not leave the hook armed for an unrelated run)
            self._before_fake = None
            self._nodes_on_side2 = False
            self._pipe_capture = False
        return {"fused": fused, "x": gx, "y": gy, "sd": sd, "sg": sg, "d_loss": d_loss, "g_loss": g_loss, "keep": self.keep_gradients, "reduced": reduced,
                "consts": F.constants_snapshot()}

    def _train_step_merged(self, d_latents, d_labels, real_images, g_latents, g_labels):
        hp = self.hyper_params
        head, fade = self._regime()
        fused = self._fused_ok()
        key = (head, fade is None, self.keep_gradients, fused)
        if not (fused and self._merged is not None and self._merged["key"] == key):
            self._join_updates()   # (the one-graph iteration applies a pending generator step itself, at the front of the replay)
        if fade is not None:
            if self._lerp is None:
                self._lerp = F.DeviceLerp(self.g_params.flat.device)
            self._lerp.set(fade)   # (stream-ordered before the replays below)
        d_in, g_in = (d_latents, d_labels, real_images), (g_latents, g_labels)
        M = self._merged
        if (M is None or M["key"] != key
                or any(a.shape != b.shape or a.dtype != b.dtype for a, b in zip(M["sd"] + M["sg"], d_in + g_in))):
            self._join_updates()   # (a pending generator step belongs to the graph being dropped)
            if M is not None and M["key"][:2] != key[:2]:
                F.drop_constants()   # (a new growing regime: see _run)
            self._graphs.clear()
            self._merged = None
            M = self._capture_merged(d_in, g_in, fused=fused)
            if M is None:   # (data parallel: the collectives were refused by the capture on some rank)
                d_loss = self.discriminator_step(d_latents, d_labels, real_images)
                g_loss = self.generator_step(g_latents, g_labels)
                return d_loss, g_loss
            M["key"] = key
            self._merged = M
        _copy_inputs(M["sd"] + M["sg"], list(d_in) + list(g_in))

        def armed(params):   # a no-fill graph is about to accumulate into this buffer: it must be clean (see _run)
            if not self.keep_gradients:
                if not params.grad_clean:
                    params.grad.zero_()
                params.grad_clean = False
        if M["fused"]:
            zero = not self.keep_gradients
            self.d_params.t += 1
            self.g_params.t += 1
            lr_d = self._lr_t(hp.discriminator_learning_rate, hp.discriminator_beta1, hp.discriminator_beta2, self.d_params.t)
            self._opt_scalars.set([lr_d, -1.0 if self._g_pending is None else self._g_pending])   # (stream-ordered before the replay)
            armed(self.d_params)
            if self._g_pending is None:
                armed(self.g_params)      # (no step at the front of this replay: the buffer must already be clean)
            self._g_pending = None
            M["x"].replay()
            self.d_params.grad_clean = zero
            self.g_params.grad_clean = False   # (holds the gradient of the step that is now pending)
            self._g_pending = self._lr_t(hp.generator_learning_rate, hp.generator_beta1, hp.generator_beta2, self.g_params.t)
            self.global_step += 1  # models.py:84
            self.discriminator_loss, self.generator_loss = M["d_loss"], M["g_loss"]
            return M["d_loss"], M["g_loss"]
        armed(self.d_params)
        M["x"].replay()
        self._apply(self.d_params, hp.discriminator_learning_rate, hp.discriminator_beta1, hp.discriminator_beta2, reduced=M["reduced"][0])
        armed(self.g_params)
        M["y"].replay()
        self._apply(self.g_params, hp.generator_learning_rate, hp.generator_beta1, hp.generator_beta2, reduced=M["reduced"][1])
        self.global_step += 1  # models.py:84
        self.discriminator_loss, self.generator_loss = M["d_loss"], M["g_loss"]
        return M["d_loss"], M["g_loss"]

    def _next_inputs(self):
        """One iteration's inputs (models.py:191-192: a fresh batch for each of the two runs).  StopIteration = the input ran dry."""
        if self._peeked is not None:
            out, self._peeked = self._peeked, None
            return out
        real_images, labels = self._real_batch()
        d_latents = self.fake_input_fn().to(self.dtype)
        _, g_labels = self.real_input_fn()  # the G run only consumes the labels of its batch (waveform branch is pruned)
        g_latents, g_labels = self.fake_input_fn().to(self.dtype), g_labels.to(self.dtype)
        return real_images, labels, d_latents, g_latents, g_labels

    def train_step(self):
        """models.py:191-192: one discriminator run then one generator run, fresh inputs for each."""
        real_images, labels, d_latents, g_latents, g_labels = self._next_inputs()
        self._ensure_built(d_latents, labels)
        if self.use_graphs:
            self._check_fork_runtime()
        if self._pipelined_ok():
            return self._train_step_pipelined(d_latents, labels, real_images, g_latents, g_labels)
        if self._merged_ok():
            return self._train_step_merged(d_latents, labels, real_images, g_latents, g_labels)
        d_loss = self.discriminator_step(d_latents, labels, real_images)
        g_loss = self.generator_step(g_latents, g_labels)
        return d_loss, g_loss

    def _all_ranks_have_input(self, have):
        """Data parallel: the input shards are rank-local (files[rank::world], per-record filters), so they run dry at different
        steps; a rank that stopped alone would leave the others blocked in the next all-reduce.  Every rank votes before each
        iteration and all stop together at the first "no" (the reference's single process stops at its OutOfRangeError,
        models.py:193).  Inputs that cannot run dry (`real_input_fn.finite == False`) skip the vote and its host sync."""
        if not self.distributed or not getattr(self.real_input_fn, "finite", True):
            return have
        dev = self.g_params.flat.device if self.g_params is not None else (torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu"))
        flag = torch.tensor([1 if have else 0], dtype=torch.int32, device=dev)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
        return bool(flag.item())

    def train(self, model_dir=None, config=None, total_steps=None, save_checkpoint_steps=1000, save_summary_steps=None, log_tensor_steps=100,
              log=print, save=None):
        """models.py:110 -- `train(model_dir, config, total_steps, save_checkpoint_steps, save_summary_steps, log_tensor_steps)`, the
        reference's own signature and argument order, so that gan_synth_main.py:102-109 calls it unchanged.  `config` is the reference's
        tf.ConfigProto (session / GPU-allocator options, gan_synth_main.py:91-98): nothing of it applies to this runtime, it is accepted
        and ignored.  `save_summary_steps` drives the reference's SummarySaverHook (TensorBoard images / audio, models.py:131-136): out of
        this path's scope, accepted and ignored.  `log`, `save` are additions (keyword only in practice).
        models.py:110-194 without the TF summary hooks: resume from the latest checkpoint of `model_dir` (CheckpointSaverHook /
        MonitoredSession semantics), alternate D and G runs until global_step reaches total_steps (StopAtStepHook) or the input
        runs dry (OutOfRangeError, :193), log the two losses every `log_tensor_steps` (LoggingTensorHook), checkpoint every
        `save_checkpoint_steps` and at the end.  Data parallel: EVERY rank passes `model_dir` and restores from the same file
        (weights, Adam slots, optimizer steps, global_step -- so that all ranks resume in the same growing regime); only rank 0
        writes (`save` defaults to rank == 0)."""
        from . import checkpoint
        if isinstance(model_dir, (int, float)) and not isinstance(model_dir, bool) and total_steps is None:
            model_dir, total_steps = None, model_dir   # (rounds 1-5 of this tree: train(total_steps, ...) with the count first)
        if total_steps is None:
            raise TypeError("train(): total_steps is required (models.py:110)")
        del config, save_summary_steps
        save = (self.rank == 0) if save is None else bool(save)
        have = True
        if model_dir is not None and self.g_params is None:
            # the variables exist from step 0 in the reference's graph: build them from the first batch's shapes and restore BEFORE the
            # stop condition is looked at (a finished run resumes to zero further steps, not one)
            try:
                self._peeked = self._next_inputs()
            except StopIteration:
                have = False
            if self._all_ranks_have_input(have):
                _, labels, d_latents, _, _ = self._peeked
                self._ensure_built(d_latents, labels)
                self.restored_from = checkpoint.restore(self, model_dir)
            else:
                have, self._peeked = False, None
        elif model_dir is not None:
            self.restored_from = checkpoint.restore(self, model_dir)
        last_saved = None
        while have and self.global_step < total_steps:
            try:
                inputs = self._next_inputs()
            except StopIteration:
                inputs = None
            if not self._all_ranks_have_input(inputs is not None):
                break
            self._peeked = inputs
            d_loss, g_loss = self.train_step()
            if log is not None and self.global_step % log_tensor_steps == 0:
                log(f"global_step = {self.global_step}, generator_loss = {float(g_loss):.6f}, "
                    f"discriminator_loss = {float(d_loss):.6f}")
            if model_dir is not None and save_checkpoint_steps and self.global_step % save_checkpoint_steps == 0:
                # (global_step is the same on every rank.)  The pending generator update may still need its all-reduce: EVERY rank
                # joins here, so that the rank-local save below finds nothing left to communicate -- a collective issued by rank 0
                # alone would pair with its peers' NEXT all-reduce and leave the job one collective out of step for good.
                self._join_updates()
                if save:
                    last_saved = self.global_step
                    checkpoint.save(self, model_dir)
        if self.g_params is not None:
            self._join_updates()   # (every rank: the last generator update, and the final save must not communicate either)
        if model_dir is not None and save and self.g_params is not None and last_saved != self.global_step:
            checkpoint.save(self, model_dir)

    def generate(self, *args, **kwargs):
        """Two forms.
        `generate(model_dir, config)` -- the reference's (models.py:232-250, called at gan_synth_main.py:128-131): restores the latest
        checkpoint of `model_dir` (MonitoredSession semantics: the initial weights when there is none), then YIELDS one numpy batch of
        fake waveforms [B, waveform_length] per batch of the input functions -- labels of `real_input_fn()`, latents of
        `fake_input_fn()`, as models.py:22-31 wires `fake_waveforms` -- until the input runs dry (OutOfRangeError, :249).  `config`
        (tf.ConfigProto) is accepted and ignored.
        `generate(latents, labels)` -- fake waveforms (a device tensor) for one given batch."""
        if "model_dir" in kwargs or (args and (args[0] is None or isinstance(args[0], (str, bytes)) or hasattr(args[0], "__fspath__"))):
            return self._generate_from(*args, **kwargs)
        return self._generate_batch(*args, **kwargs)

    def _generate_from(self, model_dir, config=None):
        from . import checkpoint
        del config
        restored = False
        while True:
            try:
                _, labels = self.real_input_fn()
            except StopIteration:
                return
            latents = self.fake_input_fn()
            dev = self.store.device if hasattr(self.store, "device") else labels.device
            latents, labels = latents.to(dev), labels.to(dev)
            self._ensure_built(latents.to(self.dtype), labels.to(self.dtype))
            if not restored:
                restored = True
                self.restored_from = checkpoint.restore(self, model_dir) if model_dir is not None else None
            yield self._generate_batch(latents, labels).float().cpu().numpy()

    def _generate_batch(self, latents, labels):
        """models.py:22-31 (`fake_waveforms`): fake waveforms for a batch."""
        self._join_updates()
        with torch.no_grad():
            images = self.generator(latents.to(self.dtype), labels.to(self.dtype))
        return spectral_ops.convert_images_to_waveform(images, **self.spectral_params)
