"""The communicator handle of the C-ABI (``mhimx_comm_*``) for Python hosts.

``torch.distributed`` stays the default transport of the data-parallel update (engine.sync_flat_gradient); this is the same ONE
collective on the flat gradient buffer driven through libmhimx.so directly - what a non-Python host (or a host without
torch.distributed) would call.  RCCL is dlopen'ed by the library; the 128-byte unique id is made on rank 0 and handed to the
other ranks through any side channel (here: a broadcast over an existing torch.distributed group, or passed in by the caller).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


def unique_id() -> bytes:
    buf = C.create_string_buffer(128)
    L.check(L.lib().mhimx_comm_unique_id(buf), "mhimx_comm_unique_id")
    return buf.raw


class NativeComm:
    """One RCCL communicator behind the C boundary.  ``init`` is collective: every rank calls it with the same id."""

    def __init__(self, rank: int, world: int, id_bytes: bytes | None = None, group=None, mode: int = 0):
        self.mode = int(mode)            # default form of allreduce(): 0 = ncclAllReduce, 1 = reduce-scatter + all-gather (mesh)
        if id_bytes is None:
            if world == 1:
                id_bytes = unique_id()
            else:
                import torch.distributed as dist
                box = [unique_id() if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                id_bytes = box[0]
        if len(id_bytes) != 128:
            raise L.MhimxError("NativeComm: the unique id is 128 bytes")
        self.rank, self.world = rank, world
        h = C.c_void_p()
        L.check(L.lib().mhimx_comm_init(C.byref(h), C.create_string_buffer(id_bytes, 128), rank, world), "mhimx_comm_init")
        self._h = h

    def allreduce(self, t: torch.Tensor, mode: int | None = None) -> torch.Tensor:
        """In-place fp32 SUM over the ranks on torch's current stream (enqueue only).  mode 1: reduce-scatter + all-gather."""
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise L.MhimxError("NativeComm.allreduce: a contiguous fp32 GPU tensor")
        L.check(L.lib().mhimx_comm_allreduce(self._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), C.c_void_p(t.data_ptr()),
                                             t.numel(), self.mode if mode is None else int(mode)), "mhimx_comm_allreduce")
        return t

    def close(self):
        if self._h:
            L.check(L.lib().mhimx_comm_destroy(self._h), "mhimx_comm_destroy")
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 - interpreter shutdown
            pass
