"""Tensor-level wrappers over the C ABI (one method per kernel entry point).

Inputs/outputs are torch CUDA(HIP) tensors; torch only supplies device memory and the
current stream.  4-D activations are logical NCHW with channels-last strides, which is the
[n][h][w][c] layout the kernels expect.  Every method launches HIP kernels from
libgansynth_hip.so -- nothing here computes with torch ops.

The autograd layer (functional.py) talks to the module-level `K` object; tests may swap it for
an emulation to check the autograd algebra on CPU, the product never does.
"""
import contextlib
import ctypes
import os

import numpy as np
import torch

from . import config
from . import _lib
from ._lib import GS_BF16, GS_F32

CL = torch.channels_last


def _dt(t):
    if t.dtype == torch.float32:
        return GS_F32
    if t.dtype == torch.bfloat16:
        return GS_BF16
    raise TypeError(f"unsupported activation dtype {t.dtype}")


def _act(x):
    """Make an activation tensor kernel-ready (channels-last for 4-D, contiguous otherwise)."""
    if x.dim() == 4:
        return x if x.is_contiguous(memory_format=CL) else x.contiguous(memory_format=CL)
    return x if x.is_contiguous() else x.contiguous()


def _f32c(t):
    t = t if t.dtype == torch.float32 else t.float()
    return t if t.is_contiguous() else t.contiguous()


def _empty_like_act(shape, ref):
    if len(shape) == 4:
        return torch.empty(shape, dtype=ref.dtype, device=ref.device, memory_format=CL)
    return torch.empty(shape, dtype=ref.dtype, device=ref.device)


# 1-bit leaky-relu masks (include/gansynth_hip.h, GS_ACT_WRITE_BITS / GS_ACT_LRELU_BITS): a bf16 leaky-relu conv output that a later masked
# data-gradient launch will read only for its signs carries those signs behind it in the same allocation -- numel values, then numel / 8 bytes --
# and the masked launches read 1/16 of the bytes.  A tensor "has bits" when its storage is exactly that long: whoever allocated it here wrote them.
_MASK_BITS = not config.flag("GS_NO_MASK_BITS")


def _empty_act_with_bits(shape, ref):
    n, c, h, w = shape
    numel = n * c * h * w
    flat = torch.empty((numel + numel // 16,), dtype=ref.dtype, device=ref.device)
    return flat[:numel].view(n, h, w, c).permute(0, 3, 1, 2)


def _bits_wanted(ref, co, act):
    return _MASK_BITS and act == _lib.ACT_LRELU and ref.dtype == torch.bfloat16 and co % 32 == 0


def _has_bits(t):
    return (_MASK_BITS and t.dtype == torch.bfloat16 and t.dim() == 4 and t.shape[1] % 32 == 0 and t.storage_offset() == 0
            and t.untyped_storage().nbytes() == t.numel() * 2 + t.numel() // 8)


def _mask_act(mask, mask_act):
    return _lib.ACT_LRELU_BITS if (mask_act == _lib.ACT_LRELU and _has_bits(mask)) else int(mask_act)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_POISON = config.flag("GS_DEBUG_POISON_WS")   # debugging: workspaces start as NaN bit patterns, so that a
                                                                     # kernel reading workspace it never wrote shows up as NaN


def _ws(nbytes, device):
    if _POISON:
        return torch.full((max(int(nbytes), 256),), 0xFF, dtype=torch.uint8, device=device)
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


def _rows_cols(x):
    """(rows, channels) of a channels-last activation: rows = n*h*w for 4-D, b for 2-D."""
    if x.dim() == 4:
        n, c, h, w = x.shape
        return n * h * w, c
    return x.shape[0], x.shape[1]


class HipKernels(object):
    def __init__(self):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GansynthHipError("gansynth_amd needs a HIP device (torch.cuda.is_available() is False)")
        _lib.check(self.lib.gs_init(), "gs_init")
        self._param_ranges = []   # (ptr, nbytes) of registered flat parameter buffers
        self._wcache = {}         # (weight ptr, map tag) -> (persistent workspace holding the re-laid operand, stamp)
        self._prep_tables = {}    # tuple of cache keys -> device table of GsPrepDesc rows (refresh_weights)
        self._derived = {}        # (parent weight ptr, lo, hi) -> [contiguous slice buffer, stamp, parent, lo, hi] (derived_slice)
        self._folds = None        # deferred bias-gradient folds while deferring: [(partial rows, out, nparts, c)]
        self._pending = None      # deferred weight gradients while deferring: {layer key: {out, bias, [(x, gy, with bias)]}}
        self._guarding = False    # inside stream_guard(): deferred operands are marked with the stream that finally reads them
        self._last_writer = {}    # inside stream_guard(): accumulate target -> (stream, event) of the last call that added into it
        self._early = None        # (min output pixels of a "large" layer, callback): see early_flush_rule
        self._seen = None         # while deferring: {layer key: [pairs recorded so far in this pass, out, bias]}
        self._expected = {}       # pass tag -> {layer key: (pairs of a whole pass, out, bias)}, learned at the final flush of the previous pass
        self._tag = None          # tag of the pass being deferred (defer_wgrad_reductions)
        self._complete = None     # (pred, callback, expected): see complete_rule

    # --------------------------------------------------------- prepared-weight workspaces
    def register_param_buffer(self, flat):
        """Weights living in `flat` may keep their kernel operand (re-laid, storage dtype) between calls."""
        self._param_ranges.append([flat.data_ptr(), flat.numel() * flat.element_size(), 0])

    def invalidate_weights(self, flat=None):
        """Parameter values changed (all registered buffers, or only the one `flat` lives in)."""
        ptr = None if flat is None else flat.data_ptr()
        for rng in self._param_ranges:
            if ptr is None or rng[0] <= ptr < rng[0] + rng[1]:
                rng[2] += 1

    def _weight_ws(self, w, tag, nbytes, plan=None):
        """-> (workspace, w_prepared).  A registered parameter gets one persistent workspace per conv map, reused
        (w_prepared = 1) until the parameter values change; anything else gets a transient workspace.
        `plan` = (GS_PREP_* map, ci, co, ksize, stride, dtype id) lets refresh_weights() rebuild the operand in a batch."""
        ptr = w.data_ptr()
        rng = next((r for r in self._param_ranges if r[0] <= ptr < r[0] + r[1]), None)
        if rng is None:
            return _ws(nbytes, w.device), 0
        key, stamp = (ptr, tag), (rng[2], w._version)
        ent = self._wcache.get(key)
        if ent is not None and ent[0].numel() >= nbytes:
            if ent[1] == stamp:
                return ent[0], 1
            ent[1] = stamp
            return ent[0], 0
        self._wcache[key] = [_ws(nbytes, w.device), stamp, w, plan, rng]
        self._prep_tables.clear()
        return self._wcache[key][0], 0

    def _stamp(self, w):
        ptr = w.data_ptr()
        rng = next((r for r in self._param_ranges if r[0] <= ptr < r[0] + r[1]), None)
        return (None if rng is None else rng[2], w._version)

    def derived_slice(self, w, lo, hi):
        """A persistent contiguous copy of w[:, :, lo:hi, :] (the 257-input-channel conv of the last discriminator block runs as
        two convs on slices of ONE variable): refreshed when the parent changes -- here on use, and by refresh_weights() right
        after an optimizer step -- so that a pass neither copies the slice nor re-lays its kernel operands every time, and a
        captured graph contains neither.  Returns an alias (fresh tensor object, same storage)."""
        key = (w.data_ptr(), int(lo), int(hi))
        ent = self._derived.get(key)
        stamp = self._stamp(w)
        with torch.no_grad():
            if ent is None or ent[0].shape[2] != hi - lo or ent[0].shape != w[:, :, lo:hi, :].shape:
                buf = w.detach()[:, :, lo:hi, :].contiguous()
                self.register_param_buffer(buf)
                ent = self._derived[key] = [buf, stamp, w, int(lo), int(hi)]
            elif ent[1] != stamp:
                self._refresh_derived(ent, w)
        return ent[0].detach()

    def _refresh_derived(self, ent, w):
        ent[0].copy_(w.detach()[:, :, ent[3]:ent[4], :])
        ent[1], ent[2] = self._stamp(w), w
        self.invalidate_weights(ent[0])

    def refresh_weights(self, flat=None):
        """Rebuild, in ONE launch, every stale prepared operand of the parameters living in `flat` (all registered
        buffers when None) -- called after an optimizer step instead of letting each conv re-lay its weight."""
        extra = []   # registered ranges of the derived slices refreshed here: their operands join the ONE launch below
        if self._derived:   # slices of parameters first (a copy each), then their operands together with everything else
            lo_, hi_ = (None, None) if flat is None else (flat.data_ptr(), flat.data_ptr() + flat.numel() * flat.element_size())
            with torch.no_grad():
                for ent in self._derived.values():
                    w = ent[2]
                    if (lo_ is None or lo_ <= w.data_ptr() < hi_) and ent[1] != self._stamp(w):
                        self._refresh_derived(ent, w)
                        extra.append(ent[0].data_ptr())
        ptr = None if flat is None else flat.data_ptr()

        def wanted(rng):
            return ptr is None or rng[0] <= ptr < rng[0] + rng[1] or any(rng[0] <= q < rng[0] + rng[1] for q in extra)

        stale = [k for k, e in self._wcache.items() if e[3] is not None and wanted(e[4]) and e[1] != (e[4][2], e[2]._version)]
        if not stale:
            return 0
        tkey = tuple(stale)
        table = self._prep_tables.get(tkey)
        if table is None:
            rows = np.zeros((len(stale), 5), dtype=np.int64)   # GsPrepDesc: 2 pointers + 6 int32 = 40 bytes
            for i, k in enumerate(stale):
                buf, _, w, plan, _ = self._wcache[k]
                rows[i, 0], rows[i, 1] = w.data_ptr(), buf.data_ptr()
                rows[i, 2:5] = np.array(plan, dtype=np.int32).view(np.int64)
            table = torch.from_numpy(rows).to(self._wcache[stale[0]][0].device)
            self._prep_tables[tkey] = table
        _lib.check(self.lib.gs_weight_prep_batch(table.data_ptr(), len(stale), _stream()), "gs_weight_prep_batch")
        for k in stale:
            e = self._wcache[k]
            e[1] = (e[4][2], e[2]._version)
        return len(stale)

    # ------------------------------------------------- accumulate targets on more than one stream
    @contextlib.contextmanager
    def _adds_into(self, *targets):
        """Around a launch that ADDS into `targets` (a variable's gradient) right away.  Inside stream_guard() two passes on two streams may
        add into the same variable -- the real and the fake pass of a discriminator run: the launch waits for the last one that touched the
        target from the other stream and leaves its own event behind (the host's issue order, which is also the order the one-stream
        schedule uses).  Calls that only RECORD a deferred job, or leave partial rows for the batched fold, do not come through here: the
        R1 double-backward's first recorded layer must not wait for the end of the fake pass's backward."""
        if not self._guarding:
            yield
            return
        stream = torch.cuda.current_stream()
        targets = [t for t in targets if isinstance(t, torch.Tensor) and t.is_cuda]
        spans = [self._span(t) for t in targets]
        for lo, hi in spans:
            # (by ADDRESS RANGE, not by pointer: a channel slice w.grad[:, :, lo:hi, :] of a wider variable -- wgrad_slice_target_ok -- and its
            #  parent, or two slices, are the same memory under different pointers)
            for (plo, phi), (pstream, pev) in self._last_writer.items():
                if plo < hi and lo < phi and pstream != stream.cuda_stream:
                    stream.wait_event(pev)
        yield
        for span in spans:
            ev = torch.cuda.Event()
            ev.record(stream)
            for old in [k for k in self._last_writer if k[0] < span[1] and span[0] < k[1] and k != span]:
                # an overlapping older entry stays only where it reaches beyond the new one (its event still orders that part)
                if span[0] <= old[0] and old[1] <= span[1]:
                    del self._last_writer[old]
            self._last_writer[span] = (stream.cuda_stream, ev)

    @staticmethod
    def _span(t):
        """[first byte, one past the last byte) a (possibly strided) tensor touches."""
        lo = t.data_ptr()
        return lo, lo + (sum((n - 1) * st for n, st in zip(t.shape, t.stride()) if n > 0) + 1) * t.element_size()

    # ----------------------------------------------------------- deferred weight gradients
    def defer_wgrad_reductions(self, tag=None):
        """`tag`: names the KIND of pass (the trainer's "d" / "g") -- the pairs each layer receives in a pass are remembered per tag (complete_rule).
        From now on the in-place (`out=`) conv weight gradients are only RECORDED; flush_wgrad_reductions() then runs, per
        weight, ONE multi-source launch over all recorded (x, gy) pairs of that layer (real + fake discriminator pass, the
        second-order contribution of the penalty terms ...) and folds the slice partials of all layers in a handful of launches.
        A backward pass has ~70 such gradients, each otherwise its own partials + reduction.  The `out` buffers are complete
        only after the flush; x and gy are kept alive (and must not be written) until then."""
        if self._pending is None:
            self._pending = {}
            self._folds = None if config.value("GS_NO_DEFERRED_FOLDS") else []
            self._seen, self._tag, self._complete = {}, tag, None

    def _partial_rows(self, producer, p, c, dt, out):
        """While gradients are deferred: a buffer of its own for the partial rows of a bias gradient that is ADDED into `out` (a
        variable's .grad, only read after the flush) -- the producer then skips its fold and flush_wgrad_reductions() folds all
        of them in one launch (gs_channel_fold_batch).  None: fold right away."""
        if self._folds is None or out is None:
            return None
        rows = self.lib.gs_bias_partial_rows(producer, p, c, dt)
        if rows <= 0:
            return None
        part = torch.empty((rows * c,), dtype=torch.float32, device=out.device)
        self._folds.append((part, out, rows, c))
        return part

    def _flush_folds(self):
        folds, self._folds = self._folds, None
        if not folds:
            return
        if self._guarding:   # (a partial written on a forked branch is read here, on the flushing stream)
            _StreamGuard._mark([f[0] for f in folds], torch.cuda.current_stream())
        arr = (_lib.GsFoldJob * len(folds))()
        for jb, (part, out, rows, c) in zip(arr, folds):
            jb.part, jb.out, jb.nparts, jb.c, jb.accumulate = part.data_ptr(), out.data_ptr(), rows, c, 1
        ptr = ctypes.cast(arr, ctypes.c_void_p)
        ws = _ws(max(self.lib.gs_channel_fold_batch_workspace_bytes(ptr, len(folds)), 256), folds[0][1].device)
        _lib.check(self.lib.gs_channel_fold_batch(ptr, len(folds), ws.data_ptr(), ws.numel(), _stream()), "gs_channel_fold_batch")

    def wgrad_slice_target_ok(self, x, co, ksize, stride):
        """May a conv weight gradient be added into a CHANNEL SLICE w.grad[:, :, lo:hi, :] of a wider variable (a strided `out`)?
        Only while gradients are deferred, for the layers gs_conv_wgrad_jobs runs grouped (bf16, 3x3, >= 64 channels both sides) and
        for the 1-input-channel plane conv of the direct kernel."""
        if self._pending is None or config.value("GS_NO_WGRAD_GROUPS") or ksize != 3:
            return False
        if x.shape[1] == 1 and stride == 1 and co % 4 == 0:   # the 1-channel direct kernel: its slice reduction stays pending, the batched fold takes the stride
            return True
        return x.dtype == torch.bfloat16 and x.shape[1] % 64 == 0 and co % 64 == 0

    # Large layers first.  A backward pass walks the pyramid top-down (and the second-order passes walk it bottom-up first): by the time it
    # reaches the few-block levels every (x, gy) pair of the full-chip levels recorded so far is final, and their contraction -- the
    # HBM-bound part of the weight gradients -- needs nothing the chain below still computes.  Rule: when a SMALL layer is recorded while
    # LARGE ones are pending, `callback(select)` is called once; the trainer answers with flush_wgrad_reductions(select=select) on a forked
    # branch of the run's hipGraph (models.GANSynth._early_flush), where it runs beside the latency-bound chain instead of after it.  A layer
    # whose pairs arrive on both sides of that point (the discriminator's: the R1 pairs early, the real / fake pairs late) is contracted in
    # two launches that add into the same gradient; the rule depends on the recorded sequence only, never on the stream.
    # Complete layers early.  The pairs a layer receives in a pass of a given kind are the same every time (the real pass, the fake pass, the
    # second-order terms): once every layer `pred` picks has received as many as in the previous pass of that kind, their gradients need nothing
    # the backward still computes -- `callback(select, others)` is called once, from inside the node that recorded the last pair, and the trainer
    # answers with flush_wgrad_reductions(select=select) and whatever wants COMPLETE gradients early (data parallel: their all-reduce, beside
    # the rest of the backward).  `others`: [(out, bias)] of every layer of the pass `pred` does not pick -- what is NOT complete then.
    def complete_rule(self, pred, callback):
        exp = self._expected.get(self._tag) if self._tag is not None else None
        self._complete = (pred, callback, exp) if (callback is not None and exp and any(pred(k) for k in exp)) else None
        return self._complete is not None

    def flush_bias_folds(self):
        """The bias-gradient folds recorded so far, now (the rest stay deferred to the final flush)."""
        if self._pending is not None and self._folds:
            self._flush_folds()
            self._folds = []

    def early_flush_rule(self, min_pixels, callback):
        """`min_pixels`: one threshold or several (descending): each fires once per arming, for the layers at or above it."""
        if callback is None:
            self._early = None
            return
        ts = sorted({int(t) for t in (min_pixels if isinstance(min_pixels, (tuple, list)) else (min_pixels,))}, reverse=True)
        self._early = (ts, callback, set())

    @staticmethod
    def _key_pixels(key):
        return int(key[6][1]) * int(key[6][2])   # (output positions of the conv whose weight this is)

    def _defer_wgrad(self, key, x, gy, out, bias_out):
        if self._early is not None and self._pending:
            px = self._key_pixels(key)
            for big in self._early[0]:
                if px < big and big not in self._early[2] and any(self._key_pixels(k) >= big for k in self._pending):
                    self._early[2].add(big)
                    self._early[1](lambda k, big=big: self._key_pixels(k) >= big)
        grp = self._pending.setdefault(key, {"out": out, "bias": None, "src": []})
        if bias_out is not None:
            assert grp["bias"] is None or grp["bias"].data_ptr() == bias_out.data_ptr()
            grp["bias"] = bias_out
        grp["src"].append((x, gy, bias_out is not None))
        if self._seen is not None:
            rec = self._seen.setdefault(key, [0, out, bias_out])
            rec[0] += 1
            if bias_out is not None:
                rec[2] = bias_out
            if self._complete is not None and self._complete[0](key):
                pred, callback, exp = self._complete
                if all(self._seen.get(k, (0,))[0] >= e[0] for k, e in exp.items() if pred(k)):
                    self._complete = None
                    callback(pred, [(e[1], e[2]) for k, e in exp.items() if not pred(k)])

    def flush_wgrad_reductions(self, group_of=None, on_group_done=None, select=None, split_stream=None, on_rest_done=None):
        """`group_of(out.data_ptr()) -> int | None` orders the layers into groups (the trainer's gradient buckets, in completion
        order); each group is contracted and folded before the next one starts and `on_group_done(group)` is called right after
        its last launch -- the data-parallel trainer puts that bucket's all-reduce on the wire there."""
        if select is not None:   # (the recorded layers `select(key)` picks, now; everything else -- and the bias folds -- stays pending)
            picked = {k: self._pending.pop(k) for k in [k for k in self._pending if select(k)]}
            return self._flush_groups(picked) if picked else 0
        groups, self._pending = self._pending, None
        if self._seen is not None and self._tag is not None:
            self._expected[self._tag] = {k: (v[0], v[1], v[2]) for k, v in self._seen.items()}
        self._seen, self._complete = None, None
        self._flush_folds()   # (bias gradients first: they belong to the same buckets as the weights folded below)
        if not groups:
            return 0
        if group_of is not None:
            tagged = {}
            for key, grp in groups.items():
                # a deferred layer writes its weight gradient AND (through the same launches) its bias gradient: it runs with the
                # EARLIER of their two buckets, or that one would go on the wire before the job has added into it
                tagged.setdefault(completion_group(group_of, grp["out"], grp["bias"]), []).append((key, grp))
            order = sorted((g for g in tagged if g is not None)) + ([None] if None in tagged else [])
            n = 0
            for g in order:
                n += self._flush_groups(dict(tagged[g]))
                if g is not None and on_group_done is not None:
                    on_group_done(g)
            return n
        return self._flush_groups(groups, split_stream, on_rest_done)

    def _flush_groups(self, groups, split_stream=None, on_rest_done=None):
        """One gs_conv_wgrad_jobs call for every recorded layer of `groups`: the library groups the layers by kernel
        instantiation (one stream-K launch + one fold per group) and batches the rest.
        `split_stream` (a captured run's idle branch stream): the HBM-bound jobs -- the <= 32-input-channel layers at the top of the pyramid,
        streaming kernels -- go THERE, beside the MFMA-bound grouped contractions of the other layers on this stream: the jobs of a flush are
        independent of each other, one launch after the other on one stream each of them has the chip to itself with half of it idle."""
        if split_stream is not None and self._guarding and len(groups) > 1:
            thin = {k: g for k, g in groups.items() if int(k[5][0]) <= 32}
            rest = {k: g for k, g in groups.items() if int(k[5][0]) > 32}
            if thin and rest and on_rest_done is not None:
                # data parallel: the grouped contractions FIRST -- they complete every gradient in front of the thin layers' in the flat buffer,
                # i.e. nearly all of its bytes -- then `on_rest_done(thin)` puts that part on the wire on this stream while the thin layers are
                # contracted beside it on the branch (which starts behind the grouped launches, not beside them: the collective takes their place)
                main = torch.cuda.current_stream()
                n = self._flush_groups(rest)
                split_stream.wait_stream(main)
                on_rest_done(thin)
                with torch.cuda.stream(split_stream):
                    n += self._flush_groups(thin)
                main.wait_stream(split_stream)
                return n
            if thin and rest:
                # (a THIRD stream for the stride-2 / transposed layers' grouped launch beside the stride-1 layers': measured neutral, 5.01 / 5.01 ms)
                main = torch.cuda.current_stream()
                split_stream.wait_stream(main)
                with torch.cuda.stream(split_stream):
                    n = self._flush_groups(thin)
                n += self._flush_groups(rest)
                main.wait_stream(split_stream)
                return n
        jobs, keep = [], []
        for key, grp in groups.items():
            kind, ksize, stride, alpha = key[0], key[2], key[3], key[4]
            out, bias, src = grp["out"], grp["bias"], grp["src"]
            if self._guarding:   # (recorded on whatever stream the backward node ran on, read on the flushing stream)
                _StreamGuard._mark([(x, gy) for x, gy, _ in src], torch.cuda.current_stream())
            _, ci, h, wd = src[0][0].shape   # (the pairs of a layer may differ in their image counts)
            co = src[0][1].shape[1]
            dt = _dt(src[0][0])
            for i in range(0, len(src), _lib.WGRAD_MAX_SOURCES):
                part = src[i:i + _lib.WGRAD_MAX_SOURCES]
                jb = _lib.GsWgradJob()
                for s_, (x, gy, wb) in enumerate(part):
                    jb.x[s_], jb.gy[s_], jb.n[s_] = x.data_ptr(), gy.data_ptr(), x.shape[0]
                jb.nsrc = len(part)
                jb.bias_mask = sum(1 << j for j, p in enumerate(part) if p[2]) if kind == "conv" else 0
                jb.gw = out.data_ptr()
                jb.gb = bias.data_ptr() if (kind == "conv" and bias is not None and jb.bias_mask) else None
                jb.h, jb.w, jb.ci, jb.co, jb.ksize, jb.stride = h, wd, ci, co, ksize, stride
                jb.transposed = 0 if kind == "conv" else 1
                jb.alpha, jb.accumulate, jb.dtype = float(alpha), 1, dt
                if not out.is_contiguous():   # [:, :, lo:hi, :] of a wider variable's gradient (wgrad_slice_target_ok)
                    assert kind == "conv" and out.stride(3) == 1 and out.stride(2) == co and out.stride(1) % co == 0 and out.stride(0) == ksize * out.stride(1)
                    jb.gw_ci_stride = out.stride(1) // co
                jobs.append(jb)
                keep.append(part)
        if jobs:
            arr = (_lib.GsWgradJob * len(jobs))(*jobs)
            ptr = ctypes.cast(arr, ctypes.c_void_p)
            nb = self.lib.gs_conv_wgrad_jobs_workspace_bytes(ptr, len(jobs))
            ws = _ws(nb, groups[next(iter(groups))]["out"].device)
            _lib.check(self.lib.gs_conv_wgrad_jobs(ptr, len(jobs), ws.data_ptr(), ws.numel(), _stream()), "gs_conv_wgrad_jobs")
        return sum(len(g["src"]) for g in groups.values())   # (workspaces and sources die here: later launches are stream-ordered behind)

    # ------------------------------------------------------------------------------- conv
    def conv2d_fwd(self, x, w, ksize, stride, alpha):
        return self.conv2d_fwd_bias_act(x, w, None, ksize, stride, alpha, _lib.ACT_NONE)

    def conv2d_fwd_mask(self, x, w, ksize, stride, alpha, mask, mask_act):
        """conv2d_fwd(x, w) * mask_act'(.) through `mask` (an activation output of the result's shape) in one pass."""
        x, w, mask = _act(x), _f32c(w), _act(mask)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        y = _empty_like_act((n, co, h // stride, wd // stride), x)
        assert mask.shape == y.shape and mask.dtype == y.dtype
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, ksize, stride, _dt(x))
        ws, prepared = self._weight_ws(w, ("fwd", ksize, stride, _dt(x)), nb, (_lib.PREP_CONV_FWD, ci, co, ksize, stride, _dt(x)))
        _lib.check(self.lib.gs_conv2d_fwd_mask(x.data_ptr(), w.data_ptr(), mask.data_ptr(), _mask_act(mask, mask_act), y.data_ptr(), n, h, wd, ci, co, ksize, stride,
                                               float(alpha), _dt(x), prepared, ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_fwd_mask")
        return y

    def conv2d_fwd_bias_act(self, x, w, bias, ksize, stride, alpha, act, bits=None):
        """`bits` (default: whenever the result qualifies, 1/16 more bytes written): the leaky-relu result carries its sign bits behind it."""
        x, w = _act(x), _f32c(w)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        bits = _bits_wanted(x, co, act) and bits is not False
        y = (_empty_act_with_bits if bits else _empty_like_act)((n, co, h // stride, wd // stride), x)
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, ksize, stride, _dt(x))
        ws, prepared = self._weight_ws(w, ("fwd", ksize, stride, _dt(x)), nb, (_lib.PREP_CONV_FWD, ci, co, ksize, stride, _dt(x)))
        bp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        _lib.check(self.lib.gs_conv2d_fwd_bias_act(x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), n, h, wd, ci, co, ksize, stride,
                                                   float(alpha), act | (_lib.ACT_WRITE_BITS if bits else 0), _dt(x), prepared, ws.data_ptr(), ws.numel(), _stream()),
                   "gs_conv2d_fwd_bias_act")
        return y

    def conv2d_fwd_bias_act_norm(self, x, w, bias, ksize, stride, alpha, act, eps, want_z=True):
        """(z, y): z = act(alpha * conv + bias), y = pixel_norm(z) -- one launch where the conv tile owns all channels of a pixel;
        z is None with want_z=False (no backward will need the activation)."""
        x, w = _act(x), _f32c(w)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        y = _empty_like_act((n, co, h // stride, wd // stride), x)
        z = _empty_like_act((n, co, h // stride, wd // stride), x) if want_z else None
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, ksize, stride, _dt(x))
        ws, prepared = self._weight_ws(w, ("fwd", ksize, stride, _dt(x)), nb, (_lib.PREP_CONV_FWD, ci, co, ksize, stride, _dt(x)))
        bp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        _lib.check(self.lib.gs_conv2d_fwd_bias_act_norm(x.data_ptr(), w.data_ptr(), bp, None if z is None else z.data_ptr(), y.data_ptr(), n, h, wd,
                                                        ci, co, ksize, stride, float(alpha), act, float(eps), _dt(x), prepared, ws.data_ptr(),
                                                        ws.numel(), _stream()), "gs_conv2d_fwd_bias_act_norm")
        return z, y

    def conv2d_transpose_fwd_bias_act_norm(self, x, w, bias, alpha, act, eps, want_z=True):
        x, w = _act(x), _f32c(w)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        y = _empty_like_act((n, co, 2 * h, 2 * wd), x)
        z = _empty_like_act((n, co, 2 * h, 2 * wd), x) if want_z else None
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, _dt(x))
        ws, prepared = self._weight_ws(w, ("t_fwd", _dt(x)), nb, (_lib.PREP_CONVT_FWD, ci, co, 3, 2, _dt(x)))
        bp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        _lib.check(self.lib.gs_conv2d_transpose_s2_fwd_bias_act_norm(x.data_ptr(), w.data_ptr(), bp, None if z is None else z.data_ptr(), y.data_ptr(),
                                                                     n, h, wd, ci, co, float(alpha), act, float(eps), _dt(x), prepared,
                                                                     ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_transpose_s2_fwd_bias_act_norm")
        return z, y

    def conv2d_bwd_data(self, gy, w, x_shape, ksize, stride, alpha, mask=None, mask_act=0):
        """gx, or with `mask` (the conv's forward input, itself the output of activation `mask_act`) gx * act'(.): the data
        gradient w.r.t. the previous layer's pre-activation in one pass."""
        gy, w = _act(gy), _f32c(w)
        n, ci, h, wd = x_shape
        co = w.shape[3]
        gx = _empty_like_act((n, ci, h, wd), gy)
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_DATA, n, h, wd, ci, co, ksize, stride, _dt(gy))
        ws, prepared = self._weight_ws(w, ("bwd_data", ksize, stride, _dt(gy)), nb, (_lib.PREP_CONV_BWD_DATA, ci, co, ksize, stride, _dt(gy)))
        mp = None
        if mask is not None:
            mask = _act(mask)
            assert mask.shape == gx.shape and mask.dtype == gx.dtype
            mp = mask.data_ptr()
        _lib.check(self.lib.gs_conv2d_bwd_data_mask(gy.data_ptr(), w.data_ptr(), mp, _mask_act(mask, mask_act) if mask is not None else 0, gx.data_ptr(), n, h, wd, ci, co, ksize, stride,
                                                    float(alpha), _dt(gy), prepared, ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_bwd_data_mask")
        return gx

    def fwd_pnbwdbwd_is_fused(self, x_shape, co, ksize, stride, transposed, dtype):
        """Does conv2d[_transpose]_fwd_pnbwdbwd run as ONE launch for a conv with input x_shape and `co` output channels?"""
        n, ci, h, wd = x_shape
        return bool(self.lib.gs_conv2d_fwd_pnbwdbwd_is_fused(int(n), int(h), int(wd), int(ci), int(co), int(ksize), int(stride), 1 if transposed else 0,
                                                            GS_F32 if dtype == torch.float32 else GS_BF16))

    def conv2d_fwd_pnbwdbwd(self, x, w, ksize, stride, alpha, g, z, eps, act):
        """(out_z, out_g) = both gradients of u = act'(z) pixel_norm_bwd(g, z) contracted with t = conv2d_fwd(x, w): what pixel_norm_bwd_bwd(t, g, z,
        pre_act=act, with_g=True) returns, from the conv's epilogue where its tile owns all channels of a pixel."""
        x, w, g, z = _act(x), _f32c(w), _act(g), _act(z)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        out_g = _empty_like_act((n, co, h // stride, wd // stride), x)
        assert g.shape == out_g.shape == z.shape and g.dtype == x.dtype == z.dtype
        out_z = torch.empty_like(out_g)
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, ksize, stride, _dt(x))
        ws, prepared = self._weight_ws(w, ("fwd", ksize, stride, _dt(x)), nb, (_lib.PREP_CONV_FWD, ci, co, ksize, stride, _dt(x)))
        _lib.check(self.lib.gs_conv2d_fwd_pnbwdbwd(x.data_ptr(), w.data_ptr(), g.data_ptr(), z.data_ptr(), int(act), float(eps), out_g.data_ptr(), out_z.data_ptr(), n, h,
                                                   wd, ci, co, ksize, stride, float(alpha), _dt(x), prepared, ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_fwd_pnbwdbwd")
        return out_z, out_g

    def conv2d_transpose_fwd_pnbwdbwd(self, x, w, alpha, g, z, eps, act):
        x, w, g, z = _act(x), _f32c(w), _act(g), _act(z)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        out_g = _empty_like_act((n, co, 2 * h, 2 * wd), x)
        assert g.shape == out_g.shape == z.shape and g.dtype == x.dtype == z.dtype
        out_z = torch.empty_like(out_g)
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, _dt(x))
        ws, prepared = self._weight_ws(w, ("t_fwd", _dt(x)), nb, (_lib.PREP_CONVT_FWD, ci, co, 3, 2, _dt(x)))
        _lib.check(self.lib.gs_conv2d_transpose_s2_fwd_pnbwdbwd(x.data_ptr(), w.data_ptr(), g.data_ptr(), z.data_ptr(), int(act), float(eps), out_g.data_ptr(),
                                                                out_z.data_ptr(), n, h, wd, ci, co, float(alpha), _dt(x), prepared, ws.data_ptr(), ws.numel(), _stream()),
                   "gs_conv2d_transpose_s2_fwd_pnbwdbwd")
        return out_z, out_g

    def bwd_data_pnbwd_is_fused(self, x_shape, co, ksize, stride, transposed, dtype):
        """Does conv2d[_transpose]_bwd_data_pnbwd run as ONE launch for a conv with input x_shape = (n, ci, h, w) and `co` output channels?"""
        n, ci, h, wd = x_shape
        return bool(self.lib.gs_conv2d_bwd_data_pnbwd_is_fused(int(n), int(h), int(wd), int(ci), int(co), int(ksize), int(stride), 1 if transposed else 0,
                                                              GS_F32 if dtype == torch.float32 else GS_BF16))

    def conv2d_bwd_data_pnbwd(self, gy, w, x_shape, ksize, stride, alpha, z, eps, act, addend=None):
        """(pixel_norm_bwd(conv2d_bwd_data(gy, w), z) + addend) * act'(z): the data gradient continued through the previous block's pixel norm
        and activation (z: that block's activation output, x_shape's shape) -- one launch where the conv tile owns all channels of a pixel."""
        gy, w, z = _act(gy), _f32c(w), _act(z)
        n, ci, h, wd = x_shape
        co = w.shape[3]
        assert tuple(z.shape) == (n, ci, h, wd) and z.dtype == gy.dtype
        gx = _empty_like_act((n, ci, h, wd), gy)
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_DATA, n, h, wd, ci, co, ksize, stride, _dt(gy))
        ws, prepared = self._weight_ws(w, ("bwd_data", ksize, stride, _dt(gy)), nb, (_lib.PREP_CONV_BWD_DATA, ci, co, ksize, stride, _dt(gy)))
        ap = None
        if addend is not None:
            addend = _match(addend, z)
            ap = addend.data_ptr()
        _lib.check(self.lib.gs_conv2d_bwd_data_pnbwd(gy.data_ptr(), w.data_ptr(), z.data_ptr(), ap, int(act), float(eps), gx.data_ptr(), n, h, wd, ci, co, ksize,
                                                     stride, float(alpha), _dt(gy), prepared, ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_bwd_data_pnbwd")
        return gx

    def conv2d_transpose_bwd_data_pnbwd(self, gy, w, alpha, z, eps, act, addend=None):
        """The same behind a transposed conv (its data gradient is the stride-2 conv)."""
        gy, w, z = _act(gy), _f32c(w), _act(z)
        n, co, h2, w2 = gy.shape
        ci = w.shape[2]
        h, wd = h2 // 2, w2 // 2
        assert tuple(z.shape) == (n, ci, h, wd) and z.dtype == gy.dtype
        gx = _empty_like_act((n, ci, h, wd), gy)
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_BWD_DATA, n, h, wd, ci, co, _dt(gy))
        ws, prepared = self._weight_ws(w, ("t_bwd_data", _dt(gy)), nb, (_lib.PREP_CONVT_BWD_DATA, ci, co, 3, 2, _dt(gy)))
        ap = None
        if addend is not None:
            addend = _match(addend, z)
            ap = addend.data_ptr()
        _lib.check(self.lib.gs_conv2d_transpose_s2_bwd_data_pnbwd(gy.data_ptr(), w.data_ptr(), z.data_ptr(), ap, int(act), float(eps), gx.data_ptr(), n, h, wd, ci, co,
                                                                  float(alpha), _dt(gy), prepared, ws.data_ptr(), ws.numel(), _stream()),
                   "gs_conv2d_transpose_s2_bwd_data_pnbwd")
        return gx

    def conv2d_bwd_weight(self, x, gy, ksize, stride, alpha, out=None, bias_out=None):
        """gw (new tensor), or with `out` (fp32, contiguous) the gradient is ADDED into it inside the kernel.  With `bias_out`
        (needs `out`) the bias gradient sum_{n,h,w} gy is added into it by the same launches."""
        x, gy = _act(x), _act(gy)
        n, ci, h, wd = x.shape
        co = gy.shape[1]
        gw = torch.empty((ksize, ksize, ci, co), dtype=torch.float32, device=x.device) if out is None else out
        if bias_out is not None:
            assert out is not None and bias_out.dtype == torch.float32 and bias_out.is_contiguous()
        if out is not None and self._pending is not None:
            self._defer_wgrad(("conv", out.data_ptr(), ksize, stride, float(alpha), tuple(x.shape[1:]), tuple(gy.shape[1:]), x.dtype), x, gy, out, bias_out)
            return gw
        assert out is None or out.is_contiguous(), "a channel-slice target needs deferred gradients (wgrad_slice_target_ok)"
        nb = self.lib.gs_conv2d_workspace_bytes(_lib.CONV_BWD_WEIGHT, n, h, wd, ci, co, ksize, stride, _dt(x))
        ws = _ws(nb, x.device)
        with self._adds_into(out, bias_out):
            _lib.check(self.lib.gs_conv2d_bwd_weight_bias(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), None if bias_out is None else bias_out.data_ptr(),
                                                          n, h, wd, ci, co, ksize, stride, float(alpha), 0 if out is None else 1, _dt(x),
                                                          ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_bwd_weight_bias")
        return gw

    def conv2d_transpose_fwd(self, x, w, alpha):
        return self.conv2d_transpose_fwd_bias_act(x, w, None, alpha, _lib.ACT_NONE)

    def conv2d_transpose_fwd_bias_act(self, x, w, bias, alpha, act):
        x, w = _act(x), _f32c(w)
        n, ci, h, wd = x.shape
        co = w.shape[3]
        y = _empty_like_act((n, co, 2 * h, 2 * wd), x)
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_FWD, n, h, wd, ci, co, _dt(x))
        ws, prepared = self._weight_ws(w, ("t_fwd", _dt(x)), nb, (_lib.PREP_CONVT_FWD, ci, co, 3, 2, _dt(x)))
        bp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        _lib.check(self.lib.gs_conv2d_transpose_s2_fwd_bias_act(x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), n, h, wd, ci, co,
                                                                float(alpha), act, _dt(x), prepared, ws.data_ptr(), ws.numel(), _stream()),
                   "gs_conv2d_transpose_s2_fwd_bias_act")
        return y

    def conv2d_transpose_bwd_data(self, gy, w, alpha):
        gy, w = _act(gy), _f32c(w)
        n, co, h2, w2 = gy.shape
        ci = w.shape[2]
        h, wd = h2 // 2, w2 // 2
        gx = _empty_like_act((n, ci, h, wd), gy)
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_BWD_DATA, n, h, wd, ci, co, _dt(gy))
        ws, prepared = self._weight_ws(w, ("t_bwd_data", _dt(gy)), nb, (_lib.PREP_CONVT_BWD_DATA, ci, co, 3, 2, _dt(gy)))
        _lib.check(self.lib.gs_conv2d_transpose_s2_bwd_data(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), n, h, wd, ci, co, float(alpha),
                                                            _dt(gy), prepared, ws.data_ptr(), ws.numel(), _stream()), "gs_conv2d_transpose_s2_bwd_data")
        return gx

    def conv2d_transpose_bwd_weight(self, x, gy, alpha, out=None):
        x, gy = _act(x), _act(gy)
        n, ci, h, wd = x.shape
        co = gy.shape[1]
        gw = torch.empty((3, 3, ci, co), dtype=torch.float32, device=x.device) if out is None else out
        if out is not None and self._pending is not None:
            self._defer_wgrad(("convT", out.data_ptr(), 3, 2, float(alpha), tuple(x.shape[1:]), tuple(gy.shape[1:]), x.dtype), x, gy, out, None)
            return gw
        nb = self.lib.gs_conv2d_transpose_s2_workspace_bytes(_lib.CONV_BWD_WEIGHT, n, h, wd, ci, co, _dt(x))
        ws = _ws(nb, x.device)
        with self._adds_into(out):
            _lib.check(self.lib.gs_conv2d_transpose_s2_bwd_weight(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), n, h, wd, ci, co, float(alpha),
                                                                  0 if out is None else 1, _dt(x), ws.data_ptr(), ws.numel(), _stream()),
                       "gs_conv2d_transpose_s2_bwd_weight")
        return gw

    # ------------------------------------------------------------------------------ dense
    def dense_fwd(self, x, w, alpha):
        x, w = _act(x), _f32c(w)
        b, i = x.shape
        o = w.shape[1]
        y = torch.empty((b, o), dtype=x.dtype, device=x.device)
        ws = _ws(self.lib.gs_dense_fwd_workspace_bytes(b, i, o), x.device)
        _lib.check(self.lib.gs_dense_fwd(x.data_ptr(), w.data_ptr(), y.data_ptr(), b, i, o, float(alpha), _dt(x),
                                         ws.data_ptr(), ws.numel(), _stream()), "gs_dense_fwd")
        return y

    def dense_fwd_bias_act(self, x, w, bias, alpha, act):
        """act(alpha * x @ w + bias): bias / activation where the forward writes its result (x 2-D, or a channels-last 4-D activation
        whose NCHW flatten feeds the layer)."""
        x, w = _act(x), _f32c(w)
        o = w.shape[1]
        y = torch.empty((x.shape[0], o), dtype=x.dtype, device=x.device)
        bp = None if bias is None else _f32c(bias).data_ptr()
        if x.dim() == 4:
            b, c, h, wd = x.shape
            ws = _ws(self.lib.gs_dense_fwd_workspace_bytes(b, c * h * wd, o), x.device)
            _lib.check(self.lib.gs_dense_fwd_bias_act_nhwc(x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), b, c, h * wd, o, float(alpha), int(act), _dt(x),
                                                           ws.data_ptr(), ws.numel(), _stream()), "gs_dense_fwd_bias_act_nhwc")
        else:
            b, i = x.shape
            ws = _ws(self.lib.gs_dense_fwd_workspace_bytes(b, i, o), x.device)
            _lib.check(self.lib.gs_dense_fwd_bias_act(x.data_ptr(), w.data_ptr(), bp, y.data_ptr(), b, i, o, float(alpha), int(act), _dt(x),
                                                      ws.data_ptr(), ws.numel(), _stream()), "gs_dense_fwd_bias_act")
        return y

    def dense_bwd_data(self, gy, w, alpha):
        gy, w = _act(gy), _f32c(w)
        b, o = gy.shape
        i = w.shape[0]
        gx = torch.empty((b, i), dtype=gy.dtype, device=gy.device)
        _lib.check(self.lib.gs_dense_bwd_data(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), b, i, o, float(alpha), _dt(gy), _stream()),
                   "gs_dense_bwd_data")
        return gx

    def drop_deferred(self):
        """Forget every deferred job (a backward pass that raised in the middle of a capture: models.GANSynth._abandon_capture)."""
        self._pending, self._folds = None, None
        self._seen, self._complete = None, None

    # (dense weight gradients deferred to the final contraction like the convs' -- off the backward's chain, beside the MFMA-bound jobs: measured
    #  neutral, 5.083 -> 5.09 ms, round 6; they are launched where autograd produces them)
    def dense_bwd_weight(self, x, gy, alpha, out=None):
        x, gy = _act(x), _act(gy)
        b, i = x.shape
        o = gy.shape[1]
        gw = torch.empty((i, o), dtype=torch.float32, device=x.device) if out is None else out
        with self._adds_into(out):
            _lib.check(self.lib.gs_dense_bwd_weight(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), b, i, o, float(alpha),
                                                    0 if out is None else 1, _dt(x), _stream()), "gs_dense_bwd_weight")
        return gw

    # the dense layer behind tf.layers.flatten of an NCHW activation (networks.py:185-186), fed with the channels-last activation
    # itself: x is [n, c, h, w] in channels-last memory, w stays [c * h * w, out] in the reference's row order
    @staticmethod
    def dense_nhwc_ok(x, out, batch_for_weight=None):
        if config.value("GS_NO_DENSE_NHWC"):   # measurement knob: the flatten copy + the plain kernels
            return False
        return x.dim() == 4 and x.is_contiguous(memory_format=CL) and out % 256 == 0 and (x.shape[1] * x.shape[2] * x.shape[3]) % 4 == 0 and x.shape[0] <= 16

    def dense_fwd_nhwc(self, x, w, alpha):
        x, w = _act(x), _f32c(w)
        b, c, h, wd = x.shape
        o = w.shape[1]
        y = torch.empty((b, o), dtype=x.dtype, device=x.device)
        ws = _ws(self.lib.gs_dense_fwd_workspace_bytes(b, c * h * wd, o), x.device)
        _lib.check(self.lib.gs_dense_fwd_nhwc(x.data_ptr(), w.data_ptr(), y.data_ptr(), b, c, h * wd, o, float(alpha), _dt(x),
                                              ws.data_ptr(), ws.numel(), _stream()), "gs_dense_fwd_nhwc")
        return y

    def dense_bwd_data_nhwc(self, gy, w, x_shape, alpha):
        gy, w = _act(gy), _f32c(w)
        b, c, h, wd = x_shape
        o = gy.shape[1]
        gx = torch.empty((b, c, h, wd), dtype=gy.dtype, device=gy.device, memory_format=CL)
        _lib.check(self.lib.gs_dense_bwd_data_nhwc(gy.data_ptr(), w.data_ptr(), gx.data_ptr(), b, c, h * wd, o, float(alpha), _dt(gy), _stream()),
                   "gs_dense_bwd_data_nhwc")
        return gx

    def dense_bwd_weight_nhwc(self, x, gy, alpha, out=None):
        x, gy = _act(x), _act(gy)
        b, c, h, wd = x.shape
        o = gy.shape[1]
        gw = torch.empty((c * h * wd, o), dtype=torch.float32, device=x.device) if out is None else out
        with self._adds_into(out):
            _lib.check(self.lib.gs_dense_bwd_weight_nhwc(x.data_ptr(), gy.data_ptr(), gw.data_ptr(), b, c, h * wd, o, float(alpha),
                                                         0 if out is None else 1, _dt(x), _stream()), "gs_dense_bwd_weight_nhwc")
        return gw

    def embedding_fwd(self, idx, w, alpha, dtype):
        w = _f32c(w)
        idx = idx.contiguous()
        b = idx.shape[0]
        rows, units = w.shape
        y = torch.empty((b, units), dtype=dtype, device=w.device)
        _lib.check(self.lib.gs_embedding_fwd(idx.data_ptr(), w.data_ptr(), y.data_ptr(), b, rows, units, float(alpha), _dt(y), _stream()),
                   "gs_embedding_fwd")
        return y

    def embedding_onehot_fwd(self, labels, w, alpha):
        """(y, idx): the rows of w selected by the first maximum of each one-hot row, and those indices (for embedding_bwd)."""
        w = _f32c(w)
        labels = labels.contiguous()
        b, rows = labels.shape
        units = w.shape[1]
        assert rows == w.shape[0]
        y = torch.empty((b, units), dtype=labels.dtype, device=w.device)
        idx = torch.empty((b,), dtype=torch.int64, device=w.device)
        _lib.check(self.lib.gs_embedding_onehot_fwd(labels.data_ptr(), w.data_ptr(), y.data_ptr(), idx.data_ptr(), b, rows, units, float(alpha), _dt(labels),
                                                    _stream()), "gs_embedding_onehot_fwd")
        return y, idx

    def embedding_bwd(self, idx, gy, rows, alpha):
        gy = _act(gy)
        idx = idx.contiguous()
        b, units = gy.shape
        gw = torch.empty((rows, units), dtype=torch.float32, device=gy.device)
        _lib.check(self.lib.gs_embedding_bwd(idx.data_ptr(), gy.data_ptr(), gw.data_ptr(), b, rows, units, float(alpha), _dt(gy), _stream()),
                   "gs_embedding_bwd")
        return gw

    # ------------------------------------------------------------------ bias / activations
    def bias_act_fwd(self, x, bias, act):
        x = _act(x)
        p, c = _rows_cols(x)
        y = torch.empty_like(x)
        bp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        _lib.check(self.lib.gs_bias_act_fwd(x.data_ptr(), bp, y.data_ptr(), p, c, act, _dt(x), _stream()), "gs_bias_act_fwd")
        return y

    # the generator's first block: dense units (channel-major) <-> channels-last activation, in the bias / activation pass (gansynth_hip.h)
    def units_bias_act_to_nhwc(self, y, bias, c, h, w, act, mask=None):
        y = _act(y)
        n = y.shape[0]
        assert y.dim() == 2 and y.shape[1] == c * h * w
        z = torch.empty((n, c, h, w), dtype=y.dtype, device=y.device, memory_format=CL)
        bp = mp = None
        if bias is not None:
            bias = _f32c(bias)
            bp = bias.data_ptr()
        if mask is not None:
            mask = _match(mask, z)
            mp = mask.data_ptr()
        _lib.check(self.lib.gs_units_bias_act_to_nhwc(y.data_ptr(), bp, mp, z.data_ptr(), n, c, h * w, int(act), _dt(y), _stream()), "gs_units_bias_act_to_nhwc")
        return z

    def nhwc_act_bwd_to_units(self, g, z, act):
        z = _act(z)
        g = _match(g, z)
        n, c, h, w = z.shape
        gu = torch.empty((n, c * h * w), dtype=z.dtype, device=z.device)
        _lib.check(self.lib.gs_nhwc_act_bwd_to_units(g.data_ptr(), z.data_ptr(), gu.data_ptr(), n, c, h * w, int(act), _dt(z), _stream()), "gs_nhwc_act_bwd_to_units")
        return gu

    def act_bwd(self, g, y, act):
        y = _act(y)
        g = _match(g, y)
        gx = torch.empty_like(y)
        _lib.check(self.lib.gs_act_bwd(g.data_ptr(), y.data_ptr(), gx.data_ptr(), y.numel(), act, _dt(y), _stream()), "gs_act_bwd")
        return gx

    def act_bwd_bias(self, g, y, act, out=None):
        """(gx, gb): activation backward and the bias gradient in one pass (gb added into `out` when given)."""
        y = _act(y)
        g = _match(g, y)
        p, c = _rows_cols(y)
        gx = torch.empty_like(y)
        gb = torch.empty((c,), dtype=torch.float32, device=y.device) if out is None else out
        part = self._partial_rows(_lib.BIAS_FROM_ACT_BWD, p, c, _dt(y), out)
        ws = part if part is not None else _ws(self.lib.gs_channel_sum_workspace_bytes(p, c), y.device)
        flags = (0 if out is None else 1) | (_lib.SUM_PARTIALS if part is not None else 0)
        with self._adds_into(out if part is None else None):   # (partial rows left for the batched fold: `out` is not touched now)
            _lib.check(self.lib.gs_act_bwd_bias(g.data_ptr(), y.data_ptr(), gx.data_ptr(), gb.data_ptr(), p, c, act, flags,
                                                _dt(y), ws.data_ptr(), ws.numel() * ws.element_size(), _stream()), "gs_act_bwd_bias")
        return gx, gb

    def tanh_bwd_bwd(self, gg, g, y):
        y = _act(y)
        gg, g = _match(gg, y), _match(g, y)
        out = torch.empty_like(y)
        _lib.check(self.lib.gs_tanh_bwd_bwd(gg.data_ptr(), g.data_ptr(), y.data_ptr(), out.data_ptr(), y.numel(), _dt(y), _stream()),
                   "gs_tanh_bwd_bwd")
        return out

    def channel_sum(self, g, out=None):
        g = _act(g)
        p, c = _rows_cols(g)
        res = torch.empty((c,), dtype=torch.float32, device=g.device) if out is None else out
        part = self._partial_rows(_lib.BIAS_FROM_CHANNEL_SUM, p, c, _dt(g), out)
        ws = part if part is not None else _ws(self.lib.gs_channel_sum_workspace_bytes(p, c), g.device)
        flags = (0 if out is None else 1) | (_lib.SUM_PARTIALS if part is not None else 0)
        with self._adds_into(out if part is None else None):
            _lib.check(self.lib.gs_channel_sum(g.data_ptr(), res.data_ptr(), p, c, flags, _dt(g), ws.data_ptr(), ws.numel() * ws.element_size(),
                                               _stream()), "gs_channel_sum")
        return res

    def pixel_norm_fwd(self, x, eps):
        x = _act(x)
        p, c = _rows_cols(x)
        y = torch.empty_like(x)
        _lib.check(self.lib.gs_pixel_norm_fwd(x.data_ptr(), y.data_ptr(), p, c, float(eps), _dt(x), _stream()), "gs_pixel_norm_fwd")
        return y

    norm_bwd_sums_bias = True   # pixel_norm_bwd(bias_out=...) exists (functional._ConvBiasActNorm)

    def norm_bwd_bias_ok(self, c, dtype=torch.float32):
        """Can pixel_norm_bwd(bias_out=...) take rows of `c` channels?  (gs_pixel_norm_bwd_fused_bias: a power of two in 4..1024 --
        the library's own answer: gs_bias_partial_rows is 0 for the shapes that kernel rejects.)"""
        return self.lib.gs_bias_partial_rows(_lib.BIAS_FROM_PIXEL_NORM_BWD, 1, int(c), GS_F32 if dtype == torch.float32 else GS_BF16) > 0

    def pixel_norm_bwd(self, g, x, eps, act=0, pre_act=0, addend=None, bias_out=None):
        """gx = (pixel_norm_bwd(g * pre_act'(x), x) + addend) * act'(x)   (x: an activation output; see gs_pixel_norm_bwd_fused).
        `bias_out` (fp32 [c], contiguous): the same pass adds sum_pixels gx into it."""
        x = _act(x)
        g = _match(g, x)
        p, c = _rows_cols(x)
        gx = torch.empty_like(x)
        ap = None
        if addend is not None:
            addend = _match(addend, x)
            ap = addend.data_ptr()
        if bias_out is not None:
            assert bias_out.dtype == torch.float32 and bias_out.is_contiguous() and bias_out.numel() == c
            part = self._partial_rows(_lib.BIAS_FROM_PIXEL_NORM_BWD, p, c, _dt(x), bias_out)
            ws = part if part is not None else _ws(self.lib.gs_pixel_norm_bwd_bias_workspace_bytes(p, c, _dt(x)), x.device)
            with self._adds_into(bias_out if part is None else None):
                _lib.check(self.lib.gs_pixel_norm_bwd_fused_bias(g.data_ptr(), x.data_ptr(), ap, gx.data_ptr(), bias_out.data_ptr(), p, c, float(eps), int(pre_act),
                                                                 int(act), 1 | (_lib.SUM_PARTIALS if part is not None else 0), _dt(x), ws.data_ptr(),
                                                                 ws.numel() * ws.element_size(), _stream()), "gs_pixel_norm_bwd_fused_bias")
            return gx
        _lib.check(self.lib.gs_pixel_norm_bwd_fused(g.data_ptr(), x.data_ptr(), ap, gx.data_ptr(), p, c, float(eps), int(pre_act), int(act), _dt(x),
                                                    _stream()), "gs_pixel_norm_bwd_fused")
        return gx

    def pixel_norm_bwd_bwd(self, gg, g, x, eps, pre_act=0, with_g=False):
        """d<gg', pixel_norm_bwd(g, x)>/dx with gg' = gg * pre_act'(x); with_g: also pixel_norm_bwd(gg', x) (same pass) -> (out, out_g)."""
        x = _act(x)
        gg, g = _match(gg, x), _match(g, x)
        p, c = _rows_cols(x)
        out = torch.empty_like(x)
        out_g = torch.empty_like(x) if with_g else None
        _lib.check(self.lib.gs_pixel_norm_bwd_bwd_fused(gg.data_ptr(), g.data_ptr(), x.data_ptr(), out.data_ptr(), None if out_g is None else out_g.data_ptr(),
                                                        p, c, float(eps), int(pre_act), _dt(x), _stream()), "gs_pixel_norm_bwd_bwd_fused")
        return (out, out_g) if with_g else out

    # --------------------------------------------------------------------- up / down scale
    def upscale2d(self, x, fy, fx, scale):
        x = _act(x)
        n, c, h, w = x.shape
        y = _empty_like_act((n, c, h * fy, w * fx), x)
        _lib.check(self.lib.gs_upscale2d(x.data_ptr(), y.data_ptr(), n, h, w, c, fy, fx, float(scale), _dt(x), _stream()), "gs_upscale2d")
        return y

    def blocksum2d(self, x, fy, fx, scale):
        x = _act(x)
        n, c, h, w = x.shape
        y = _empty_like_act((n, c, h // fy, w // fx), x)
        _lib.check(self.lib.gs_blocksum2d(x.data_ptr(), y.data_ptr(), n, h, w, c, fy, fx, float(scale), _dt(x), _stream()), "gs_blocksum2d")
        return y

    # ------------------------------------------------------------------------ batch stddev
    # (`sub`: x is `sub` batches concatenated along axis 0 -- the statistic of ops.py:336-348 within each: one call per sub-batch on its slab)
    def batch_stddev_fwd(self, x, eps, sub=1):
        x = _act(x)
        n, c, h, w = x.shape
        y = _empty_like_act((n, 1, h, w), x)
        m, es = n // sub, x.element_size()
        for s in range(sub):
            _lib.check(self.lib.gs_batch_stddev_fwd(x.data_ptr() + s * m * c * h * w * es, y.data_ptr() + s * m * h * w * es, m, h * w, c, float(eps), _dt(x),
                                                    _stream()), "gs_batch_stddev_fwd")
        return y

    def batch_stddev_bwd(self, gy, x, eps, addend=None, sub=1):
        """d batch_stddev(x) / d x applied to gy, plus `addend` (another gradient into x) in the same pass."""
        x = _act(x)
        n, c, h, w = x.shape
        gy = _act(gy.to(x.dtype))
        gx = torch.empty_like(x)
        if addend is not None:
            addend = _match(addend, x)
        m, es = n // sub, x.element_size()
        for s in range(sub):
            ox, oy = s * m * c * h * w * es, s * m * h * w * es
            _lib.check(self.lib.gs_batch_stddev_bwd(gy.data_ptr() + oy, x.data_ptr() + ox, None if addend is None else addend.data_ptr() + ox, gx.data_ptr() + ox,
                                                    m, h * w, c, float(eps), _dt(x), _stream()), "gs_batch_stddev_bwd")
        return gx

    def batch_stddev_bwd_bwd(self, ggx, gy, x, eps, sub=1):
        x = _act(x)
        n, c, h, w = x.shape
        ggx = _match(ggx, x)
        gy = _act(gy.to(x.dtype))
        ggy = torch.empty_like(gy)
        gx2 = torch.empty_like(x)
        m, es = n // sub, x.element_size()
        for s in range(sub):
            ox, oy = s * m * c * h * w * es, s * m * h * w * es
            _lib.check(self.lib.gs_batch_stddev_bwd_bwd(ggx.data_ptr() + ox, gy.data_ptr() + oy, x.data_ptr() + ox, ggy.data_ptr() + oy, gx2.data_ptr() + ox,
                                                        m, h * w, c, float(eps), _dt(x), _stream()), "gs_batch_stddev_bwd_bwd")
        return ggy, gx2

    # ------------------------------------------------------------------- misc elementwise
    def axpby(self, a, b, ca, cb):
        a = _act(a)
        b = _match(b, a)
        out = torch.empty_like(a)
        if hasattr(ca, "owner") or hasattr(cb, "owner"):   # functional.DeviceLerp.Coef: coefficients read from a device table
            if not (hasattr(ca, "owner") and hasattr(cb, "owner") and ca.owner is cb.owner):
                raise TypeError("axpby: device coefficients must come from one DeviceLerp table")
            _lib.check(self.lib.gs_axpby_dev(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), ca.owner.table.data_ptr(), ca.index, cb.index,
                                             _dt(a), _stream()), "gs_axpby_dev")
            return out
        _lib.check(self.lib.gs_axpby(a.data_ptr(), b.data_ptr(), out.data_ptr(), a.numel(), float(ca), float(cb), _dt(a), _stream()), "gs_axpby")
        return out

    def sumsq_rows(self, x):
        x = _act(x)
        rows = x.shape[0]
        out = torch.empty((rows,), dtype=torch.float32, device=x.device)
        ws = _ws(self.lib.gs_sumsq_rows_workspace_bytes(rows), x.device)
        _lib.check(self.lib.gs_sumsq_rows(x.data_ptr(), out.data_ptr(), rows, x.numel() // rows, _dt(x), ws.data_ptr(), ws.numel(), _stream()), "gs_sumsq_rows")
        return out

    def row_scale(self, x, s, alpha=1.0):
        """out[r] = alpha * s[r] * x[r]."""
        x = _act(x)
        s = _f32c(s)
        rows = x.shape[0]
        out = torch.empty_like(x)
        _lib.check(self.lib.gs_row_scale(x.data_ptr(), s.data_ptr(), float(alpha), out.data_ptr(), rows, x.numel() // rows, _dt(x), _stream()), "gs_row_scale")
        return out

    def gan_d_loss(self, real_logits, fake_logits, labels, penalty, penalty_weight=1.0, out=None):
        """(loss, g_real_logits, g_fake_logits, g_penalty): mean of softplus(-r) + softplus(f) + penalty_weight * penalty and its gradients, one launch.
        Either logits tensor may be None (the other half of the sum is then another launch: see include/gansynth_hip.h)."""
        real_logits = None if real_logits is None else _act(real_logits)
        fake_logits = None if fake_logits is None else _act(fake_logits)
        ref = real_logits if real_logits is not None else fake_logits
        labels = _match(labels, ref)
        n, c = ref.shape
        loss = torch.empty((), dtype=torch.float32, device=ref.device)
        if out is None:
            g_real = None if real_logits is None else torch.empty_like(real_logits)
            g_fake = None if fake_logits is None else torch.empty_like(fake_logits)
        else:
            g_real, g_fake = out
        pp = g_pen = None
        if penalty is not None:
            penalty = _f32c(penalty)
            pp = penalty.data_ptr()
            g_pen = torch.empty((n,), dtype=torch.float32, device=ref.device)
        ptr = lambda t: None if t is None else t.data_ptr()
        _lib.check(self.lib.gs_gan_d_loss(ptr(real_logits), ptr(fake_logits), labels.data_ptr(), pp, float(penalty_weight), n, c, loss.data_ptr(),
                                          ptr(g_real), ptr(g_fake), ptr(g_pen), _dt(ref), _stream()), "gs_gan_d_loss")
        return loss, g_real, g_fake, g_pen

    def gan_g_loss(self, fake_logits, labels, sumsq, weight, eps):
        """(loss, g_fake_logits, g_sumsq): mean of softplus(-f) + weight / (sumsq + eps) and its gradients, one launch.  `fake_logits` None:
        the mode-seeking half alone."""
        sp = gp = None
        g_sumsq = None
        if sumsq is not None:
            sumsq = _f32c(sumsq)
            g_sumsq = torch.empty_like(sumsq)
            sp, gp = sumsq.data_ptr(), g_sumsq.data_ptr()
        if fake_logits is None:
            loss = torch.empty((), dtype=torch.float32, device=sumsq.device)
            _lib.check(self.lib.gs_gan_g_loss(None, None, sp, float(weight), float(eps), sumsq.numel(), 1, loss.data_ptr(), None, gp, _lib.GS_F32, _stream()),
                       "gs_gan_g_loss")
            return loss, None, g_sumsq
        fake_logits = _act(fake_logits)
        labels = _match(labels, fake_logits)
        n, c = fake_logits.shape
        loss = torch.empty((), dtype=torch.float32, device=fake_logits.device)
        g_fake = torch.empty_like(fake_logits)
        _lib.check(self.lib.gs_gan_g_loss(fake_logits.data_ptr(), labels.data_ptr(), sp, float(weight), float(eps), n, c, loss.data_ptr(), g_fake.data_ptr(), gp,
                                          _dt(fake_logits), _stream()), "gs_gan_g_loss")
        return loss, g_fake, g_sumsq

    def adam_tf_step(self, p, g, m, v, lr_t, beta1, beta2, eps, grad_scale=1.0, refresh=True, zero_grad=False):
        """`refresh=False`: the caller updates a buffer range by range (gradient buckets) and refreshes the operands once.
        `zero_grad`: g is cleared behind the update (the next run accumulates from zero without a fill pass)."""
        for t in (p, g, m, v):
            assert t.dtype == torch.float32 and t.is_contiguous()
        fn = self.lib.gs_adam_tf_step_zero_grad if zero_grad else self.lib.gs_adam_tf_step
        _lib.check(fn(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr_t), float(beta1),
                      float(beta2), float(eps), float(grad_scale), _stream()), "gs_adam_tf_step")
        if refresh:
            self.invalidate_weights(p)  # parameter values changed: cached kernel operands of that buffer are stale ...
            self.refresh_weights(p)     # ... and are rebuilt here in one launch

    def adam_tf_step_dev(self, p, g, m, v, lr_t_ptr, beta1, beta2, eps, grad_scale=1.0, refresh=True, zero_grad=False):
        """adam_tf_step with lr_t read from device memory when the launch EXECUTES (`lr_t_ptr`: address of one fp32; negative = no step):
        the form a captured graph can hold.  `refresh`: the prepared operands of every weight in `p` are rebuilt behind it, unconditionally
        -- inside a capture the host-side staleness stamps describe the capture pass, not the replays."""
        for t in (p, g, m, v):
            assert t.dtype == torch.float32 and t.is_contiguous()
        _lib.check(self.lib.gs_adam_tf_step_dev(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), int(lr_t_ptr), float(beta1),
                                                float(beta2), float(eps), float(grad_scale), 1 if zero_grad else 0, _stream()), "gs_adam_tf_step_dev")
        if refresh:
            self.invalidate_weights(p)
            self.refresh_weights(p)

    # ------------------------------------------------------------------- more than one stream
    def stream_guard(self):
        """`with K.stream_guard():` -- every tensor argument of every kernel-layer call made inside is marked with the stream the call
        runs on (Tensor.record_stream: a no-op for the stream the tensor was allocated on).  A run captured with forked branches
        (models.GANSynth._branch) reads tensors on a stream other than their own; torch's caching allocator would otherwise hand a freed
        block straight back to ITS stream while the other stream's kernel is still reading it.  Inside a stream capture a marked block
        is not reused until the capture ends.  Costs host time at capture only; the wrappers exist while the context does."""
        return _StreamGuard(self)

    # --------------------------------------------------------------------------- profiling
    def account(self):
        """bench.py: `with K.account() as calls:` lists every kernel-layer call made inside as (method, argument dict, bytes read,
        bytes written) -- the algorithmic traffic of SURVEY.md 8(d): every tensor argument read once, every result written once
        (`out=` targets that are accumulated into: read and written).  Measurement only: the wrappers exist while the context does."""
        return _Accounting(self)

    def prof_enable(self, on):
        """on: False / True, or an int n > 1 for burst mode (every conv launch n times back to back inside its event pair)."""
        self.lib.gs_prof_enable(int(on))

    def prof_roofline(self, peak_tflops, peak_gbps):
        """(algorithmic bytes, roofline ms, HBM-bound part of it) of the recorded launches; call before prof_collect()."""
        b, r, rh = ctypes.c_double(0.0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        self.lib.gs_prof_roofline(float(peak_tflops), float(peak_gbps), ctypes.byref(b), ctypes.byref(r), ctypes.byref(rh))
        return b.value, r.value, rh.value

    def prof_records(self, max_records=8192):
        """[(ms, flops, bytes, (kind, N, Hb, Wb, IC, OC, a, b))] of the recorded launches; call before prof_collect()."""
        n = ctypes.c_int(0)
        ms = (ctypes.c_double * max_records)()
        fl = (ctypes.c_double * max_records)()
        by = (ctypes.c_double * max_records)()
        desc = (ctypes.c_int * (8 * max_records))()
        self.lib.gs_prof_records(max_records, ctypes.byref(n), ms, fl, by, desc)
        return [(ms[i], fl[i], by[i], tuple(desc[8 * i:8 * i + 8])) for i in range(n.value)]

    def prof_collect(self):
        n, ms, fl = ctypes.c_int(0), ctypes.c_double(0.0), ctypes.c_double(0.0)
        self.lib.gs_prof_collect(ctypes.byref(n), ctypes.byref(ms), ctypes.byref(fl))
        return n.value, ms.value, fl.value


class _Accounting(object):
    SKIP = ("account", "prof_enable", "prof_roofline", "prof_records", "prof_collect", "register_param_buffer", "invalidate_weights", "early_flush_rule",
            "derived_slice", "defer_wgrad_reductions", "wgrad_slice_target_ok", "drop_deferred", "dense_nhwc_ok", "norm_bwd_bias_ok",
            "fwd_pnbwdbwd_is_fused", "bwd_data_pnbwd_is_fused")

    def __init__(self, K):
        self.K, self.calls, self.depth, self.names = K, [], 0, []

    @staticmethod
    def _nbytes(obj):
        if isinstance(obj, torch.Tensor):
            return obj.numel() * obj.element_size()
        if isinstance(obj, (tuple, list)):
            return sum(_Accounting._nbytes(o) for o in obj)
        return 0

    def _wrap(self, name, fn):
        import inspect
        sig = inspect.signature(fn)

        def wrapper(*a, **kw):
            self.depth += 1
            try:
                out = fn(*a, **kw)
            finally:
                self.depth -= 1
            if self.depth == 0:   # (conv2d_fwd -> conv2d_fwd_bias_act ...: the outermost call is the one counted)
                try:
                    args = dict(sig.bind(*a, **kw).arguments)
                except TypeError:
                    args = {}
                read = sum(self._nbytes(v) for k, v in args.items() if k not in ("out", "bias_out"))
                acc = sum(self._nbytes(args.get(k)) for k in ("out", "bias_out"))   # accumulated into: read + written
                written = self._nbytes(out) if acc == 0 or not isinstance(out, torch.Tensor) else 0
                meta = {k: (tuple(v.shape) if isinstance(v, torch.Tensor) else (tuple(v) if isinstance(v, (tuple, list, torch.Size)) else v))
                        for k, v in args.items()
                        if isinstance(v, (int, float, bool, torch.Tensor)) or v is None
                        or (isinstance(v, (tuple, list, torch.Size)) and all(isinstance(e, int) for e in v))}
                self.calls.append((name, meta, read + acc, written + acc))
            return out
        return wrapper

    def __enter__(self):
        for name in dir(type(self.K)):
            if name.startswith("_") or name in self.SKIP:
                continue
            fn = getattr(self.K, name)
            if callable(fn):
                setattr(self.K, name, self._wrap(name, fn))   # (instance attribute shadowing the method)
                self.names.append(name)
        return self.calls

    def __exit__(self, *exc):
        for name in self.names:
            try:
                delattr(self.K, name)
            except AttributeError:
                pass
        return False


class _StreamGuard(object):
    SKIP = _Accounting.SKIP + ("stream_guard", "refresh_weights", "flush_wgrad_reductions")

    hook = None   # class attribute: `hook(wrapper, name) -> wrapper` (debugging scripts only)

    def __init__(self, K):
        self.K, self.names = K, []

    @staticmethod
    def _mark(obj, stream):
        if isinstance(obj, torch.Tensor):
            if obj.is_cuda:
                obj.record_stream(stream)
        elif isinstance(obj, (tuple, list)):
            for o in obj:
                _StreamGuard._mark(o, stream)

    def _wrap(self, fn, name=""):
        mark = self._mark
        hook = self.hook   # (scripts/dbg_fork2.py: a recording wrapper around every call; None in production)

        def wrapper(*a, **kw):
            stream = torch.cuda.current_stream()
            for v in a:
                mark(v, stream)
            for v in kw.values():
                mark(v, stream)
            return fn(*a, **kw)
        return wrapper if hook is None else hook(wrapper, name)

    def __enter__(self):
        for name in dir(type(self.K)):
            if name.startswith("_") or name in self.SKIP or name in self.K.__dict__:
                continue
            fn = getattr(self.K, name)
            if callable(fn):
                setattr(self.K, name, self._wrap(fn, name))   # (instance attribute shadowing the method)
                self.names.append(name)
        self.K._guarding = True
        self.K._last_writer = {}
        return self

    def __exit__(self, *exc):
        self.K._guarding = False
        self.K._last_writer = {}
        for name in self.names:
            try:
                delattr(self.K, name)
            except AttributeError:
                pass
        return False


def completion_group(group_of, *targets):
    """The group (gradient bucket, in completion order) a deferred job runs in when it adds into several `targets` (a layer's weight
    gradient and, from the same launches, its bias gradient): the EARLIEST of their groups.  Buckets go on the wire in index order,
    bucket i right after the jobs of group i -- so a job must have run by the time the first bucket it touches is sent (the later
    ones are sent after it anyway).  (Also used by the CPU emulation of the tests.)"""
    tags = [group_of(t.data_ptr()) for t in targets if t is not None]
    tags = [t for t in tags if t is not None]
    return min(tags) if tags else None


def _match(t, ref):
    """Bring a gradient tensor to the dtype/layout of the activation it pairs with."""
    if t.dtype != ref.dtype:
        t = t.to(ref.dtype)
    if t.shape != ref.shape:
        t = t.expand(ref.shape)
    return _act(t)


_K = None


def get():
    """The process-wide kernel object (created on first use; raises without the library / a GPU)."""
    global _K
    if _K is None:
        _K = HipKernels()
    return _K


def set_backend(obj):
    """Test hook: install an object with the HipKernels interface (CPU emulation for autograd tests)."""
    global _K
    _K = obj
