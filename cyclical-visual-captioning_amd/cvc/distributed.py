"""One process per GPU, clips sharded across ranks, ONE exchange per training step: the gradient
all-reduce (RCCL over xGMI; backend "nccl" is RCCL on ROCm, "gloo" for the CPU tests).

The reference's only multi-GPU mechanism is single-process nn.DataParallel (main.py:169): scatter
the batch, per-replica token-mean losses, unweighted mean over replicas, gradients reduced to one
device.  The process-per-GPU equivalent is sum-all-reduce of gradients divided by the world size
(SURVEY.md section 8(e)), before clip_grad_norm_ and the optimizer step (trainer.py:118-122).

Design for xGMI (7 point-to-point links per GPU, no switch): few, large messages.  Gradients are
packed into flat buckets (default 128 MB) in REVERSE registration order -- the order backward
produces them: the vocabulary head first, the LSTM cells (whose gradients complete only when BPTT
reaches t = 0) last -- and each bucket's all-reduce is launched asynchronously from a
post-accumulate-grad hook as soon as its last gradient lands, so the logit/embed traffic overlaps
the recurrent backward.  Parameters whose gradient is None (the dead i2h_2 / h2h_2 / localied_fc /
reconstructor soft_attn parameters) are skipped on every rank alike; a
rank-agreement check guards against divergent None-ness.
"""
from __future__ import annotations

import os
from typing import Dict, Iterable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: str = "nccl") -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_range(n: int, rank: int, world: int, equal: bool = False) -> slice:
    """Contiguous split of n clips; the first n % world ranks get one extra clip.  equal=True drops the remainder
    instead, so that every rank holds exactly n // world clips (same number of steps and collectives on every rank)."""
    base, extra = divmod(n, world)
    if equal:
        return slice(rank * base, (rank + 1) * base)
    start = rank * base + min(rank, extra)
    return slice(start, start + base + (1 if rank < extra else 0))


def shard_batch(batch, rank: int, world: int):
    """Slice every batch-leading tensor (or list) of a batch tuple/dict along dim 0."""
    def cut(x, sl):
        if isinstance(x, torch.Tensor) or isinstance(x, (list, tuple)) and not isinstance(x, str):
            return x[sl]
        return x
    if isinstance(batch, dict):
        n = next(v.shape[0] for v in batch.values() if isinstance(v, torch.Tensor))
        sl = shard_range(n, rank, world)
        return {k: cut(v, sl) for k, v in batch.items()}
    n = next(v.shape[0] for v in batch if isinstance(v, torch.Tensor))
    sl = shard_range(n, rank, world)
    return type(batch)(cut(v, sl) for v in batch)


def gather_eval_outputs(predictions: dict, grd_output: dict):
    """Inference has no collective in its data path (clips are sharded); the per-rank outputs are merged host-side:
    every rank contributes its {video: [segment predictions]} / {video: {segment: grounding}} dicts, all ranks receive
    the union (rank order, so the result is deterministic).  No-op without an initialised process group."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return predictions, grd_output
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, (dict(predictions), dict(grd_output)))
    preds, grd = {}, {}
    for p, g in parts:
        for vid, items in p.items():
            preds.setdefault(vid, []).extend(items)
        for vid, segs in g.items():
            grd.setdefault(vid, {}).update(segs)
    return preds, grd


class GradReducer:
    """Bucketed, hook-driven gradient averaging across ranks."""

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], bucket_mb: float = 128.0,
                 group=None, overlap: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        seen, params = set(), []
        for name, p in named_params:
            if p.requires_grad and id(p) not in seen:       # shared LSTM cells are listed once
                seen.add(id(p))
                params.append((name, p))
        params.reverse()                                    # backward order
        cap = int(bucket_mb * (1 << 20) / 4)
        self.buckets: List[List[Tuple[str, torch.nn.Parameter]]] = [[]]
        size = 0
        for name, p in params:
            if size and size + p.numel() > cap:
                self.buckets.append([])
                size = 0
            self.buckets[-1].append((name, p))
            size += p.numel()
        self._bucket_of: Dict[int, int] = {id(p): i for i, b in enumerate(self.buckets) for _, p in b}
        self._pending: List[int] = []
        self._ready: List[int] = []
        self._work = []
        self._active: Optional[List[bool]] = None           # which params carry gradients (agreed once)
        self.overlap = overlap and self.world > 1
        self._hooks = []
        if self.overlap:
            for _, p in params:
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))
        self._reset_counts()

    # ------------------------------------------------------------------
    def _reset_counts(self):
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def _expected(self, i: int) -> int:
        if self._active is None:
            return 1 << 30                                   # first step: no early launches
        return sum(1 for _, p in self.buckets[i] if self._active_by_id[id(p)])

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        self._ready[i] += 1
        if self._active is not None and not self._launched[i] and self._ready[i] >= self._expected(i):
            self._launch(i)

    def _launch(self, i: int):
        grads = [p.grad for _, p in self.buckets[i] if p.grad is not None]
        self._launched[i] = True
        if not grads:
            return
        flat = torch.cat([g.reshape(-1) for g in grads])
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._work.append((work, flat, grads))

    def _agree_on_active(self):
        """All ranks must skip the same None-grad parameters; agree once (MAX over a 0/1 mask)."""
        flags = [0 if p.grad is None else 1 for b in self.buckets for _, p in b]
        dev = next((p.grad.device for b in self.buckets for _, p in b if p.grad is not None), torch.device("cpu"))
        mask = torch.tensor(flags, dtype=torch.int32, device=dev)
        if self.world > 1:
            dist.all_reduce(mask, op=dist.ReduceOp.MAX, group=self.group)
        agreed = mask.tolist()
        k = 0
        self._active_by_id = {}
        for b in self.buckets:
            for _, p in b:
                if agreed[k] and p.grad is None:
                    p.grad = torch.zeros_like(p)             # another rank has a gradient here
                self._active_by_id[id(p)] = bool(agreed[k])
                k += 1
        self._active = [bool(a) for a in agreed]

    def finalize(self):
        """Call after backward(): launches what the hooks did not, waits, writes back averages."""
        if self.world == 1:
            self._reset_counts()
            return
        if self._active is None:
            self._agree_on_active()
        for i in range(len(self.buckets)):
            if not self._launched[i]:
                self._launch(i)
        inv = 1.0 / self.world
        for work, flat, grads in self._work:
            work.wait()
            off = 0
            for g in grads:
                n = g.numel()
                g.copy_(flat[off:off + n].view_as(g)).mul_(inv)
                off += n
        self._work.clear()
        self._reset_counts()

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
