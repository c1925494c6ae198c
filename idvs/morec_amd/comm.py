"""``MorecComm``: the C-ABI collectives (``morec_comm_*`` in ``include/morec_hip.h``: RCCL all-gather / reduce-scatter / all-reduce
enqueued on torch's CURRENT stream, i.e. stream-ordered with the kernels of the step) behind tensor arguments.

``torch.distributed`` stays the rendezvous: the 128-byte ``ncclUniqueId`` travels through the already initialised process group
(any backend), the communicator itself is the library's own.  Opt-in (``TrainStep(comm="rccl")`` / ``MOREC_COMM=rccl``): the
default data-parallel path keeps ``torch.distributed``'s collectives."""
from __future__ import annotations

import ctypes as C

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class MorecComm:
    def __init__(self, rank: int | None = None, world: int | None = None, device=None):
        L = _lib.lib()
        if not L.morec_comm_available():
            raise _lib.MorecError("librccl.so.1 not found in the process: " + L.morec_comm_last_error(None).decode())
        inited = dist.is_available() and dist.is_initialized()
        self.rank = (dist.get_rank() if inited else 0) if rank is None else rank
        self.world = (dist.get_world_size() if inited else 1) if world is None else world
        if device is not None:
            torch.cuda.set_device(device)
        torch.cuda.current_stream()          # the HIP context of this process's device must exist before ncclCommInitRank
        uid = C.create_string_buffer(128)
        if self.rank == 0:
            check(L.morec_comm_unique_id(uid), "morec_comm_unique_id")
        if self.world > 1:
            box = [bytes(uid.raw)]
            dist.broadcast_object_list(box, src=0)          # out-of-band exchange through the existing group
            uid = C.create_string_buffer(box[0], 128)
        h = C.c_void_p()
        check(L.morec_comm_create(C.byref(h), uid, self.rank, self.world), "morec_comm_create")
        self._h = h

    def _ok(self, rc, what):
        if rc != 0:
            raise _lib.MorecError(f"{what} failed: rc={rc} ({_lib.lib().morec_comm_last_error(self._h).decode()})")

    def all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """[n, ...] per rank -> [world * n, ...] (rank-major), on the current stream."""
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self._ok(_lib.lib().morec_comm_all_gather(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()),
                                                  t.numel() * t.element_size(), _stream()), "morec_comm_all_gather")
        return out

    def reduce_scatter_sum(self, t: torch.Tensor) -> torch.Tensor:
        """fp32 [world * n, ...] -> this rank's [n, ...] block of the sum over ranks."""
        assert t.dtype == torch.float32 and t.shape[0] % self.world == 0
        t = t.contiguous()
        n = t.shape[0] // self.world
        out = torch.empty((n,) + tuple(t.shape[1:]), device=t.device, dtype=t.dtype)
        self._ok(_lib.lib().morec_comm_reduce_scatter_f32(self._h, C.c_void_p(t.data_ptr()), C.c_void_p(out.data_ptr()),
                                                          out.numel(), _stream()), "morec_comm_reduce_scatter_f32")
        return out

    def all_reduce_sum_(self, t: torch.Tensor) -> torch.Tensor:
        assert t.dtype == torch.float32 and t.is_contiguous()
        self._ok(_lib.lib().morec_comm_all_reduce_f32(self._h, C.c_void_p(t.data_ptr()), t.numel(), _stream()),
                 "morec_comm_all_reduce_f32")
        return t

    def close(self):
        if self._h is not None and self._h.value:
            _lib.lib().morec_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass
