"""RCCL communicator of the data-parallel trainer (gs_comm_* of libgansynth_hip.so, include/gansynth_hip.h).

torch.distributed stays the launcher-facing layer (rendezvous, rank / world, the barrier of bench.py); the gradient all-reduce
itself goes through this communicator because it must run ON THE STREAM OF THE BACKWARD: torch's ProcessGroupNCCL runs every
collective on its own stream, and on this ROCm stack the event hop between a hipGraph replay and another stream costs ~0.25 ms
each way -- two all-reduces per iteration through it cost 0.49 ms of a 7.3 ms step on ONE GPU (profiles/r02_dist_overhead.txt).
"""
import ctypes

import torch

from . import _lib
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
        ident = (ctypes.c_ubyte * 128)()
        if self.rank == 0:
            _lib.check(self.lib.gs_comm_unique_id(ident), "gs_comm_unique_id")
        carrier = torch.tensor(list(ident), dtype=torch.uint8, device=device)
        dist.broadcast(carrier, 0)   # the 128-byte id travels over the launcher's own process group
        ident = (ctypes.c_ubyte * 128)(*carrier.cpu().tolist())
        self.handle = ctypes.c_void_p()
        with torch.cuda.device(device):
            _lib.check(self.lib.gs_comm_init(ctypes.byref(self.handle), self.rank, self.world, ident), "gs_comm_init")

    def all_reduce_(self, tensor):
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.is_cuda
        _lib.check(self.lib.gs_allreduce_sum_f32(self.handle, tensor.data_ptr(), tensor.numel(),
                                                 ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_allreduce_sum_f32")
        return _Done()

    def broadcast_(self, tensor, root=0):
        assert tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.is_cuda
        _lib.check(self.lib.gs_broadcast_f32(self.handle, tensor.data_ptr(), tensor.numel(), int(root),
                                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "gs_broadcast_f32")

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
    return RcclComm(device)
