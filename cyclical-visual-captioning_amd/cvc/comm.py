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

    def __init__(self, world: int, rank: int, uid: bytes, host_buffers: bool = False):
        """host_buffers=True is for ONE caller: the CPU tests that put a stand-in librccl.so (host pointers, POSIX shared memory,
        built by the test with gcc) in front of the loader's search path to execute the rank > 0 shard arithmetic of
        cvc_allreduce_grads without GPUs (tests/test_distributed_cpu.py).  It only relaxes THIS class's tensor checks -- which
        library dlopen finds is the process environment's business -- and is refused when a GPU is visible: the real RCCL would
        dereference host pointers on the device."""
        if len(uid) != 128:
            raise ValueError("RcclComm: the unique id is 128 bytes (cvc_comm_unique_id)")
        if host_buffers and torch.cuda.is_available():
            raise RuntimeError("RcclComm(host_buffers=True) is the CPU tests' stand-in transport; refused on a machine with a GPU")
        self.world, self.rank, self.host_buffers = int(world), int(rank), bool(host_buffers)
        self._h = C.c_void_p()
        buf = (C.c_char * 128).from_buffer_copy(uid)
        hip._check(hip.lib().cvc_comm_init(self.world, self.rank, buf, C.byref(self._h)), "cvc_comm_init")
        self.device = None if host_buffers else torch.cuda.current_device()

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
    def from_process_group(cls, group=None, host_buffers: bool = False) -> "RcclComm":
        """Every rank of an initialised torch.distributed group (any backend; "gloo" is the intended one) calls this: rank 0 draws
        the unique id, the control plane carries its 128 bytes, every rank joins.  The device must be set (torch.cuda.set_device).
        ALL ranks leave on the same side: after the join every rank tells the control plane whether IT holds a communicator; if any
        rank does not, the ranks that do destroy theirs and every rank raises -- a caller's fallback (bench.py's decode mode) is then
        taken by all ranks or by none, never by one rank alone while the others wait in a collective."""
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
        comm, err = None, None
        try:
            comm = cls(world, rank, box[0], host_buffers=host_buffers)
        except Exception as e:              # noqa: BLE001 -- handed to every rank below
            err = f"rank {rank}: {type(e).__name__}: {e}"
        errs = [None] * world
        dist.all_gather_object(errs, err, group=group)
        bad = [e for e in errs if e]
        if bad:
            if comm is not None:
                comm.destroy()
            raise RuntimeError("RcclComm.from_process_group: the communicator did not come up on every rank: " + "; ".join(bad)[:600])
        return comm

    def all_reduce_(self, flat: torch.Tensor, stream: Optional["torch.cuda.Stream"] = None) -> None:
        if self._h is None or not self._h.value:
            raise RuntimeError("RcclComm: the communicator was destroyed")
        if self.host_buffers:
            if not (flat.device.type == "cpu" and flat.dtype == torch.float32 and flat.is_contiguous()):
                raise TypeError("RcclComm(host_buffers=True).all_reduce_: a contiguous fp32 CPU tensor is required")
            hip._check(hip.lib().cvc_allreduce_grads(self._h, flat.data_ptr(), flat.numel(), None), "cvc_allreduce_grads")
            return
        if not (flat.is_cuda and flat.dtype == torch.float32 and flat.is_contiguous()):
            raise TypeError("RcclComm.all_reduce_: a contiguous fp32 GPU tensor is required")
        s = (stream if stream is not None else torch.cuda.current_stream()).cuda_stream
        hip._check(hip.lib().cvc_allreduce_grads(self._h, flat.data_ptr(), flat.numel(), s), "cvc_allreduce_grads")

    def count_ranks(self) -> int:
        """SUM of one 1.0 per rank through the communicator itself: how many ranks RCCL really connected"""
        one = torch.ones(1) if self.host_buffers else torch.ones(1, device=torch.device("cuda", self.device))
        self.all_reduce_(one)
        return int(round(float(one.item())))

    def destroy(self) -> None:
        """drains the device first: a communicator must not go away under collectives (or captured graphs being replayed)"""
        if self._h is not None and self._h.value:
            if not self.host_buffers:
                torch.cuda.synchronize()
            h, self._h = self._h, None
            hip._check(hip.lib().cvc_comm_destroy(h), "cvc_comm_destroy")
