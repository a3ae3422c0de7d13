"""The gradient exchange's own RCCL communicator (csrc/comm_rccl.hip: cvc_comm_* / cvc_allreduce_grads), one per process.

Why not c10d's "nccl" process group for the data path: c10d keeps a watchdog thread per RCCL group that polls the events of
outstanding collectives with hipEventQuery; while another thread captures a stream that query fails, the watchdog dies with the
exception and takes the rank down (round 4 met this about once in five captures, and lived with it by sleeping before a capture).
A communicator that nothing but the step's own streams ever touches has no such thread: collectives are plain stream operations
-- enqueued eagerly or recorded into the HIP graph of the training step -- ordered by HIP events.  torch.distributed stays as the
CONTROL plane only (rendezvous, the 128-byte unique id, host-side barriers and the max-over-ranks of bench.py), on "gloo": no
RCCL process group, no watchdog.

Replaces the reduction of the reference's nn.DataParallel (main.py:169) under trainer.py:116-122.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import hip


class RcclComm:
    """ncclCommInitRank on the current device.  `all_reduce_(flat fp32 tensor, stream)`: in-place SUM over the ranks as a
    reduce-scatter + all-gather pair (one all-reduce when the element count does not divide by the world size)."""

    def __init__(self, world: int, rank: int, uid: bytes):
        if len(uid) != 128:
            raise ValueError("RcclComm: the unique id is 128 bytes (cvc_comm_unique_id)")
        self.world, self.rank = int(world), int(rank)
        self._h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(uid)
        hip._check(hip.lib().cvc_comm_init(self.world, self.rank, buf, C.byref(self._h)), "cvc_comm_init")
        self.device = torch.cuda.current_device()

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_char * 128)()
        hip._check(hip.lib().cvc_comm_unique_id(buf), "cvc_comm_unique_id")
        return bytes(buf.raw)

    @classmethod
    def single(cls) -> "RcclComm":
        """one-rank communicator on the current device (tests, `bench.py --always-exchange` at N = 1): no torch.distributed"""
        return cls(1, 0, cls.unique_id())

    @classmethod
    def from_process_group(cls, group=None) -> "RcclComm":
        """Every rank of an initialised torch.distributed group (any backend; "gloo" is the intended one) calls this: rank 0 draws
        the unique id, the control plane carries its 128 bytes, every rank joins.  The device must be set (torch.cuda.set_device)."""
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            raise RuntimeError("RcclComm.from_process_group: torch.distributed is not initialised (use RcclComm.single() for one rank)")
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [None]
        if rank == 0:
            try:
                box[0] = cls.unique_id()
            except Exception as e:          # every rank must learn of it: the others are about to wait for this broadcast
                box[0] = e
        dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        if isinstance(box[0], Exception):
            raise RuntimeError(f"RcclComm.from_process_group: rank 0 could not draw the unique id: {box[0]}")
        return cls(world, rank, box[0])

    def all_reduce_(self, flat: torch.Tensor, stream: Optional["torch.cuda.Stream"] = None) -> None:
        if self._h is None or not self._h.value:
            raise RuntimeError("RcclComm: the communicator was destroyed")
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise TypeError("RcclComm.all_reduce_: a contiguous fp32 GPU tensor is required")
        s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        hip._check(hip.lib().cvc_allreduce_grads(self._h, flat.data_ptr(), flat.numel(), s), "cvc_allreduce_grads")

    def count_ranks(self) -> int:
        """SUM of one 1.0 per rank through the communicator itself: how many ranks RCCL really connected"""
        one = torch.ones(1, device=torch.device("cuda", self.device))
        self.all_reduce_(one)
        return int(round(float(one.item())))

    def destroy(self) -> None:
        """drains the device first: a communicator must not go away under collectives (or captured graphs being replayed)"""
        if self._h is not None and self._h.value:
            torch.cuda.synchronize()
            h, self._h = self._h, None
            hip._check(hip.lib().cvc_comm_destroy(h), "cvc_comm_destroy")
