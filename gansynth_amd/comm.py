"""RCCL communicator of the data-parallel trainer (gs_comm_* of libgansynth_hip.so, include/gansynth_hip.h).

torch.distributed stays the launcher-facing layer (rendezvous, rank / world, the barrier of bench.py); the gradient all-reduce
itself goes through this communicator because it must run ON THE STREAM OF THE BACKWARD: torch's ProcessGroupNCCL runs every
collective on its own stream, and on this ROCm stack the event hop between a hipGraph replay and another stream costs ~0.25 ms
each way -- two all-reduces per iteration through it cost 0.49 ms of a 7.3 ms step on ONE GPU (profiles/r02_dist_overhead.txt).
"""
import ctypes

import torch

from . import _lib
from . import config
from . import kernels


class _Done(object):
    """What torch.distributed's async ops return, for a collective that is already ordered on the current stream."""

    def wait(self):
        return True


class RcclComm(object):
    def __init__(self, device):
        import torch.distributed as dist
        self.lib = kernels.get().lib
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        # Every rank runs the same sequence of launcher-side collectives whatever fails locally (a rank that raised before its
        # peers' broadcast would leave them waiting): errors are collected, agreed on with MIN all-reduces, and raised together.
        # ncclCommInitRank BLOCKS until all `world` ranks have entered it, so the ranks agree that every one of them can (librccl
        # resolvable, rank 0 produced an id) BEFORE anyone calls it -- a rank that skipped it would hang its peers inside it.
        ident = (ctypes.c_ubyte * 128)()
        error = None
        try:
            _lib.check(self.lib.gs_comm_available(), "gs_comm_available")
            if self.rank == 0:
                _lib.check(self.lib.gs_comm_unique_id(ident), "gs_comm_unique_id")
        except Exception as e:   # noqa: BLE001 -- reported below, on every rank
            error = e
        carrier = torch.tensor(list(ident), dtype=torch.uint8, device=device)
        dist.broadcast(carrier, 0)   # the 128-byte id travels over the launcher's own process group
        ident = (ctypes.c_ubyte * 128)(*carrier.cpu().tolist())
        self.handle = ctypes.c_void_p()
        ready = torch.tensor([0 if error is not None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if int(ready.item()) == 0:
            raise RuntimeError("RCCL cannot be used by libgansynth_hip.so on every rank (rank %d: %s)"
                               % (self.rank, error if error is not None else "ok here, failed on a peer"))
        try:
            with torch.cuda.device(device):
                _lib.check(self.lib.gs_comm_init(ctypes.byref(self.handle), self.rank, self.world, ident), "gs_comm_init")
        except Exception as e:   # noqa: BLE001
            error = e
        ok = torch.tensor([0 if error is not None else 1], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            self.close()
            raise RuntimeError("RCCL communicator of libgansynth_hip.so could not be created on every rank (rank %d: %s)"
                               % (self.rank, error if error is not None else "ok here, failed on a peer"))
        marker = config.value("GS_COMM_MARKER_US")   # (tests / profiles at world size 1: read ONCE, here -- the all-reduce itself reads no environment)
        if marker is not None and self.world == 1:
            self.set_marker_us(float(marker))

    def set_marker_us(self, us):
        """Tests / profiles, world size 1 only: the all-reduce becomes a one-block kernel holding its stream for `us` microseconds."""
        _lib.check(self.lib.gs_comm_set_marker_us(self.handle, float(us)), "gs_comm_set_marker_us")
        self.marker_us = float(us)

    def all_reduce_(self, tensor, marker_share=None):
        """`marker_share` (stand-ins only): this message's share of the bytes the stand-in's time was chosen for -- a gradient sent in two
        messages must not be priced as two whole ones (15 us: what a message costs however short)."""
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.is_cuda
        full = getattr(self, "marker_us", -1.0)
        if marker_share is not None and full > 0.0:
            _lib.check(self.lib.gs_comm_set_marker_us(self.handle, max(15.0, full * float(marker_share))), "gs_comm_set_marker_us")
            try:
                _lib.check(self.lib.gs_allreduce_sum_f32(self.handle, tensor.data_ptr(), tensor.numel(),
                                                         ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_allreduce_sum_f32")
            finally:
                _lib.check(self.lib.gs_comm_set_marker_us(self.handle, full), "gs_comm_set_marker_us")
            return _Done()
        _lib.check(self.lib.gs_allreduce_sum_f32(self.handle, tensor.data_ptr(), tensor.numel(),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_allreduce_sum_f32")
        return _Done()

    def broadcast_(self, tensor, root=0):
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.is_cuda
        _lib.check(self.lib.gs_broadcast_f32(self.handle, tensor.data_ptr(), tensor.numel(), int(root),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_broadcast_f32")

    def count(self):
        """ncclCommCount of the communicator (what RCCL itself says it spans)."""
        n = ctypes.c_int(0)
        _lib.check(self.lib.gs_comm_count(self.handle, ctypes.byref(n)), "gs_comm_count")
        return int(n.value)

    def close(self):
        if self.handle:
            self.lib.gs_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


def create(device):
    """A communicator for the current torch.distributed job, or None when the job is not on HIP devices over the nccl (= RCCL)
    backend (the CPU / gloo tests keep torch.distributed's own collectives)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or torch.device(device).type != "cuda":
        return None
    if dist.get_backend() != "nccl" or not hasattr(kernels.get(), "lib") or kernels.get().lib is None:
        return None
    try:
        return RcclComm(device)
    except RuntimeError as e:
        # Agreed on by all ranks (see RcclComm.__init__).  The job continues on torch.distributed's own RCCL communicator -- the same
        # library, its collectives on the communicator's stream instead of the backward's -- and says so.
        import sys
        print("gansynth_amd.comm: %s; falling back to torch.distributed (nccl = RCCL) collectives" % e, file=sys.stderr, flush=True)
        return None
