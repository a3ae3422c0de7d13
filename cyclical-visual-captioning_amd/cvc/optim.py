"""Adam with the gradient clip folded into its step -- the optimizer side of the training path on the HIP kernels.

`ClipAdam` IS a torch.optim.Adam (same constructor arguments, param groups, `state_dict()` layout: per-parameter `step`,
`exp_avg`, `exp_avg_sq`), so checkpoints, LR schedulers and resume code see nothing new.  What changes is how a step is taken
on the GPU: `clip_and_step(max_norm, inv_world)` does `nn.utils.clip_grad_norm_` + `optimizer.step()` of the reference's
trainer (trainer.py:116-122) in three launches of csrc/optim.hip (norm partials, coefficient + step counters, one pass over
p / g / m / v) with nothing read back by the host -- the library path takes a norm reduction per gradient bucket, a multiply
over every gradient and the fused Adam's own pass.  `step()` alone is the unclipped update on the same kernel.

Parameters whose .grad is None are skipped, as torch.optim.Adam skips them (no state, no step count, no weight decay).
amsgrad / maximize / differentiable are not taken: the reference never sets them (main.py:171-191)."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import hip

_SEG_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("step", "<u8"), ("n", "<i8"), ("lr", "<f4"), ("wd", "<f4")])
_CHUNK_DT = np.dtype([("seg", "<i4"), ("pad", "<i4"), ("start", "<i8")])


class ClipAdam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, **kw):
        for k in ("amsgrad", "maximize", "differentiable"):
            if kw.pop(k, False):
                raise ValueError(f"ClipAdam: {k} is not supported")
        kw.pop("fused", None); kw.pop("foreach", None); kw.pop("capturable", None)
        # capturable=True: the step counters are float32 tensors on the parameters' device, which is what the kernel increments
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, capturable=True, foreach=False, fused=False)
        self._table = None          # (key, device segment table, chunk table, scratch, norm_coef, ...)
        self._seg_host = None       # host copy of the segment table (lr / weight_decay are rewritten in place when they change)
        self._retired = []          # tables replaced by a rebuild: kept alive, a captured HIP graph may still replay on them
        self.last_norm: Optional[torch.Tensor] = None       # device scalar: norm of the (averaged) gradient of the last clipped step

    # ------------------------------------------------------------------ state, exactly as torch.optim.Adam lays it out
    def _ensure_state(self, p):
        st = self.state[p]
        if len(st) == 0:
            st["step"] = torch.zeros((), dtype=torch.float32, device=p.device)
            st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        return st

    def _build_table(self):
        segs, keep, key = [], [], []
        betas = eps = None
        for group in self.param_groups:
            b = tuple(float(x) for x in group["betas"])
            e = float(group["eps"])
            if betas is None:
                betas, eps = b, e
            elif (betas, eps) != (b, e):
                raise ValueError("ClipAdam: all param groups must share betas and eps (they do in the reference, main.py:171-191)")
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                        and p.grad.dtype == torch.float32):
                    raise RuntimeError("ClipAdam: contiguous float32 GPU parameters and gradients only (no CPU fallback)")
                st = self._ensure_state(p)
                if not (st["step"].is_cuda and st["step"].dtype == torch.float32):      # a state_dict loaded from a CPU-step checkpoint
                    st["step"] = torch.as_tensor(float(st["step"]), dtype=torch.float32, device=p.device)
                lr = group["lr"]
                segs.append((p, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], float(lr), float(group["weight_decay"])))
                key.append((p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()))
        if not segs:
            return None
        key = (tuple(key), betas, eps)
        if self._table is not None and self._table[0] == key:
            # same tensors: only the learning rates / weight decays may have moved (a scheduler).  They live in the device table,
            # rewritten IN PLACE so that a captured graph replaying this table honours them
            lr_wd = np.array([(s_[5], s_[6]) for s_ in segs], dtype=np.float32)
            if not (np.array_equal(self._seg_host["lr"], lr_wd[:, 0]) and np.array_equal(self._seg_host["wd"], lr_wd[:, 1])):
                self._seg_host["lr"], self._seg_host["wd"] = lr_wd[:, 0], lr_wd[:, 1]
                self._table[1].copy_(torch.from_numpy(self._seg_host.view(np.uint8).copy()).pin_memory(), non_blocking=False)
            return self._table
        dev = segs[0][0].device
        chunk = int(hip.lib().cvc_optim_chunk_elems())
        # the two tables as numpy records with the C structs' layout (cvc_optim_seg: 5 pointers, int64, 2 floats; cvc_optim_chunk)
        seg_np = np.zeros(len(segs), dtype=_SEG_DT)
        assert seg_np.itemsize == C.sizeof(hip.OptimSeg) and _CHUNK_DT.itemsize == C.sizeof(hip.OptimChunk)
        for i, (p, g, m, v, step, lr, wd) in enumerate(segs):
            seg_np[i] = (p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), step.data_ptr(), p.numel(), lr, wd)
        counts = np.array([(p.numel() + chunk - 1) // chunk for p, *_ in segs], dtype=np.int64)
        chunk_np = np.zeros(int(counts.sum()), dtype=_CHUNK_DT)
        chunk_np["seg"] = np.repeat(np.arange(len(segs), dtype=np.int32), counts)
        first = np.concatenate(([0], np.cumsum(counts)[:-1]))
        chunk_np["start"] = (np.arange(len(chunk_np), dtype=np.int64) - np.repeat(first, counts)) * chunk
        seg_t = torch.from_numpy(seg_np.view(np.uint8)).to(dev)
        chunk_t = torch.from_numpy(chunk_np.view(np.uint8)).to(dev)
        partial = torch.empty(len(chunk_np), dtype=torch.float32, device=dev)
        norm_coef = torch.zeros(2, dtype=torch.float32, device=dev)
        if self._table is not None:
            self._retired.append(self._table)
        self._seg_host = seg_np
        self._table = (key, seg_t, chunk_t, partial, norm_coef, len(segs), len(chunk_np), betas, eps)
        return self._table

    def sync_hyperparameters(self):
        """Push the param groups' current lr / weight_decay into the device table (in place).  Eager steps do this on their own;
        call it before replaying a HIP graph that captured clip_and_step (Trainer.train_step_graphed does)."""
        if self._table is not None:
            self._build_table()

    # ------------------------------------------------------------------ the step
    @torch.no_grad()
    def clip_and_step(self, max_norm: float, inv_world: float = 1.0, write_grad: bool = True, zero_grad: bool = False,
                      skip: Optional[torch.Tensor] = None):
        """clip_grad_norm_(all parameters, max_norm) then step(), on the device.  inv_world = 1 / ranks when the gradients are sums
        over ranks (cvc.distributed.GradReducer.finalize(average=False)).  zero_grad: leave every gradient this step consumed
        ZERO (the next step's zero_grad() folded into the pass; instead of the clipped gradient write_grad leaves).  Returns the
        norm as a device scalar.  skip: a device int32 word (cvc.hip.step_status); non-zero when the pass executes = the step is void,
        nothing but the gradients' zeroing happens (see include/cvc_hip.h)."""
        if skip is not None and not (skip.is_cuda and skip.dtype == torch.int32 and skip.numel() >= 1):
            raise TypeError("ClipAdam.clip_and_step: skip must be an int32 tensor on the GPU")
        t = self._build_table()
        if t is None:
            return None
        _key, seg_t, chunk_t, partial, norm_coef, nseg, nchunk, betas, eps = t
        hip._check(hip.lib().cvc_adam_clip_step(seg_t.data_ptr(), nseg, chunk_t.data_ptr(), nchunk, float(max_norm), float(inv_world),
                                                betas[0], betas[1], eps, 2 if zero_grad else (1 if write_grad else 0), partial.data_ptr(),
                                                norm_coef.data_ptr(), skip.data_ptr() if skip is not None else None, hip._stream()),
                   "cvc_adam_clip_step")
        self.last_norm = norm_coef[0]
        return self.last_norm

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.clip_and_step(0.0, 1.0, write_grad=False)
        return loss
