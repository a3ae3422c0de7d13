"""Host -> HBM staging one batch ahead.

The reference moves every batch to the device synchronously at the top of the step
(trainer.py:71-82: `.cuda()` on pageable memory), which at raw-feature sizes (frame features
[B, 480, 3072] fp32 = 377 MB per 64 clips) is several milliseconds of PCIe time in front of a ~35 ms
step.  `DevicePrefetcher` issues the copies of batch k+1 on a side HIP stream from pinned memory while
step k computes, and hands the device tensors over with the stream dependency and allocator ownership
(`record_stream`) set, so the consumer uses them like any other tensor.
"""
from __future__ import annotations

from typing import Callable, Iterable, Iterator

import torch


def _walk(obj, fn):
    if isinstance(obj, torch.Tensor):
        return fn(obj)
    if isinstance(obj, dict):
        return {k: _walk(v, fn) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_walk(v, fn) for v in obj)
    return obj


def pin(batch):
    """Page-lock every CPU tensor of a (nested) batch (no-op for tensors that already are)."""
    return _walk(batch, lambda t: t if (t.is_cuda or t.is_pinned()) else t.pin_memory())


class DevicePrefetcher:
    """for prepared in DevicePrefetcher(loader, prepare, device): ...

    `prepare(batch)` is the consumer's own host->device function (e.g. Trainer._prepare); it runs under the side
    stream, so its `.to(device)` calls become asynchronous copies there.  Iteration order and contents are exactly
    the loader's."""

    def __init__(self, loader: Iterable, prepare: Callable, device: torch.device, limit: int | None = None):
        self.loader, self.prepare, self.device, self.limit = loader, prepare, device, limit
        self.stream = torch.cuda.Stream(device)

    def _stage(self, it: Iterator):
        try:
            batch = next(it)
        except StopIteration:
            return None
        with torch.cuda.stream(self.stream):
            return self.prepare(pin(batch))

    def __iter__(self):
        it = iter(self.loader)
        staged = self._stage(it)
        n = 0
        while staged is not None and (self.limit is None or n < self.limit):
            cur = torch.cuda.current_stream(self.device)
            cur.wait_stream(self.stream)
            _walk(staged, lambda t: t.record_stream(cur) if t.is_cuda else None)
            ready = staged
            n += 1
            staged = self._stage(it) if (self.limit is None or n < self.limit) else None
            yield ready
