"""Train / eval harness with the reference's hooks (trainer.py:26-373): the 12-tuple batch
contract, the per-batch trimming, the 11-tensor model call, the loss mix (ground_loss is returned
but never optimised), zero_grad -> backward -> [gradient all-reduce] -> clip_grad_norm_(0.1) ->
step.  Reference bugs are not copied: `next(it)` instead of `.next()`, the
reconstruction-loss meter is updated, grounding stats are only logged when they exist.

Caption / grounding scoring lives in external toolkits the reference vendors as (empty)
submodules; pass `scorer(predictions, grd_output, opts) -> dict` to get language stats.
"""
from __future__ import annotations

import json
import os
import time
from collections import defaultdict

import torch
import torch.nn as nn

from .distributed import GradReducer, gather_eval_outputs
from .prefetch import DevicePrefetcher
from .misc import utils
from .misc.utils import AverageMeter


def _trim(t, n, dim=1):
    return t.narrow(dim, 0, n)


SHAPE_BUCKETS = 4         # trimmed lengths are rounded up to one of this many sizes per axis (graphed training: one graph per shape)


def bucket_len(n: int, full: int, buckets: int = SHAPE_BUCKETS) -> int:
    """n rounded up to the next of `buckets` evenly spaced lengths <= full (the loader's padded length).  What lies between n and
    the bucket's length is the loader's OWN padding -- zero rows with their mask bit set (dataloader_anet.py:376-388), exactly what
    a shorter clip of the batch already carries up to the batch maximum (trainer.py:63-69) -- so the step computes the same
    function of the batch, only the set of distinct shapes becomes small enough to keep one captured graph per shape.
    ONE EXCEPTION, handled by the caller (Trainer._prepare): a clip whose proposals are ALL masked.  The reference's mask fill is the
    finite -1e8 (model/modules.py:122-129), so such a row's softmax is uniform over all N positions of the trimmed axis and its
    context the mean of N rows; a longer axis changes that N.  A batch that contains a clip with zero proposals is therefore
    trimmed exactly as the reference trims it, never bucketed."""
    if buckets <= 0 or n >= full:
        return min(n, full)
    step = -(-full // buckets)
    return min(full, -(-n // step) * step)


def _shape_key(b) -> tuple:
    out = []
    for k in sorted(b):
        v = b[k]
        if isinstance(v, torch.Tensor):
            out.append((k, tuple(v.shape), str(v.dtype)))
        elif isinstance(v, dict):
            out.append((k, tuple((kk, tuple(vv.shape), str(vv.dtype)) for kk, vv in sorted(v.items()) if isinstance(vv, torch.Tensor))))
    return tuple(out)


def _copy_into(static, b):
    for k, v in b.items():
        if isinstance(v, torch.Tensor):
            static[k].copy_(v, non_blocking=True)
        elif isinstance(v, dict):
            for kk, vv in v.items():
                if isinstance(vv, torch.Tensor):
                    static[k][kk].copy_(vv, non_blocking=True)
        else:
            static[k] = v


class Trainer:
    def __init__(self, opts, dataset, model, optimizer, train_loader, val_loader, scorer=None, grad_reducer=None):
        self.opts = opts
        self.dataset = dataset
        self.model = model
        self.optimizer = optimizer
        self.train_loader = train_loader
        self.val_loader = val_loader
        self.scorer = scorer
        self.grad_reducer = grad_reducer
        self.device = next(model.parameters()).device
        self._graph = None            # HIP-graph state of train_step_graphed
        # ---- train(): one captured graph per (bucketed) batch shape
        # shape buckets exist for ONE purpose -- a small set of shapes to keep captured graphs for -- so they are active only while
        # train() replays graphs (decided at the start of every train(): graph_capable()); an eager run (raw features through an
        # encoder that cannot be captured, a c10d / gloo exchange, a non-capturable optimizer, the CPU) trims exactly as the
        # reference does (trainer.py:63-69) and pays for no padding
        self.shape_buckets = SHAPE_BUCKETS if bool(getattr(opts, "hip_graph", 0)) else 0
        self._active_buckets = 0
        # deferred error words (cvc.hip.defer_errors): the step's status word while train() runs a model whose persistent kernels
        # report time-outs (the once-per-clip encoder's GRU) as captured / replayed steps; None otherwise
        self._status = None
        self.deferred_stats = dict(void_steps=0, rerun_steps=0)
        self._graphs = {}             # shape key -> (graph, static inputs, static result)
        self._shape_seen = {}         # shape key -> eager steps taken at that shape
        self._eager_steps = 0
        self._graph_pool = None
        self._side = None
        self._graph_broken = False
        self.graph_stats = dict(eager=0, replayed=0, captured=0)

    # ------------------------------------------------------------------ batch plumbing
    def _prepare(self, batch, train: bool):
        seg_feat, iseq, gts_seq, num, proposals, bboxs, box_mask, seg_id, region_feat, frm_mask, sample_idx, ppl_mask = batch
        n_prop = max(int(num[:, 1].max()), 1)
        # a clip with zero proposals is a uniform softmax over the trimmed axis (bucket_len's exception): trim such a batch exactly
        buckets = self._active_buckets if (train and int(num[:, 1].min()) > 0) else 0
        if buckets:
            n_prop = bucket_len(n_prop, proposals.size(1), buckets)
        proposals, ppl_mask, region_feat = _trim(proposals, n_prop), _trim(ppl_mask, n_prop), _trim(region_feat, n_prop)
        if train:
            n_box = max(int(num[:, 2].max()), 1)
            if buckets:
                n_box = bucket_len(n_box, bboxs.size(1), buckets)
            bboxs, box_mask = _trim(bboxs, n_box), _trim(box_mask, n_box, 2)
            frm_mask = _trim(_trim(frm_mask, n_prop), n_box, 2)
        dev = self.device
        # async from pinned memory; a trimmed view that already lives on the device is made dense (the kernels take contiguous tensors)
        to = lambda x: x.to(dev, non_blocking=True).contiguous() if isinstance(x, torch.Tensor) else x
        if isinstance(seg_feat, dict):
            # pre-extracted features (cvc.model.captioner.PrecomputedRegionFeatures): their region axis is trimmed with the proposals
            reg = {"pool_feats": n_prop, "p_pool_feats": n_prop, "g_pool_feats": n_prop, "pnt_mask": n_prop + 1}
            seg = {k: to(_trim(v, reg[k]) if (k in reg and v.size(1) > reg[k]) else v) for k, v in seg_feat.items()}
        else:
            seg = to(seg_feat).float()
        mask_ppls = to(ppl_mask)
        pnt_mask = torch.cat((mask_ppls.new_zeros(mask_ppls.size(0), 1), mask_ppls), dim=1)
        return dict(segs_feat=seg, input_seqs=to(iseq), gt_seqs=to(gts_seq), num=to(num), ppls=to(proposals),
                    gt_bboxs=to(bboxs), mask_bboxs=to(box_mask), ppls_feat=to(region_feat), mask_frms=to(frm_mask),
                    sample_idx=to(sample_idx).type_as(to(iseq)), pnt_mask=pnt_mask, seg_id=seg_id)

    def _call(self, b, lang_eval=False):
        return self.model(b["segs_feat"], b["input_seqs"], b["gt_seqs"], b["num"], b["ppls"], b["gt_bboxs"], b["mask_bboxs"],
                          b["ppls_feat"], b["mask_frms"], b["sample_idx"], b["pnt_mask"], lang_eval)

    def loss_mix(self, out):
        """trainer.py:92-109"""
        o = self.opts
        # the reference takes .mean() of every loss because DataParallel gathers one value per replica; one process per GPU holds
        # one value: the mean of a one-element tensor is that element (no reduction kernel, none in the backward)
        one = lambda x: x.reshape(()) if x.numel() == 1 else x.mean()
        lm_loss, att2_loss, _ground_loss, cls_loss = [one(x) for x in out[:4]]
        lm_recon = one(out[4]) if len(out) > 4 else torch.zeros((), device=lm_loss.device)
        # terms with a zero weight (w_att2 = w_cls = 0 by default, opts.py:72-75) add an exact zero and a zero gradient: left out
        terms = [(o.xe_loss_weight, lm_loss), (o.w_att2, att2_loss), (o.w_cls, cls_loss)]
        if len(out) > 4:
            terms.append((o.caption_consistency_loss_weight, lm_recon))
        loss = None
        for wgt, val in terms:
            if wgt != 0:
                loss = wgt * val if loss is None else loss + wgt * val
        if loss is None:
            loss = 0.0 * lm_loss
        return loss, lm_loss, att2_loss, cls_loss, lm_recon

    def train_step(self, batch):
        return self.train_step_prepared(self._prepare(batch, True))

    def train_step_prepared(self, b):
        """One optimisation step on a batch that already is on the device (see cvc.prefetch.DevicePrefetcher)."""
        if self.opts.att_model != 'cyclical':
            raise ValueError('Unknown att_model: {}'.format(self.opts.att_model))
        out = self._call(b)
        loss, lm, att2, cls, rec = self.loss_mix(out)
        self._backward_and_update(loss)
        self._weights_changed()
        return loss.detach(), lm.detach(), att2.detach(), cls.detach(), rec.detach()

    def _backward_and_update(self, loss, set_to_none=True):
        """zero_grad -> backward -> [gradient exchange] -> clip_grad_norm_ -> step (trainer.py:116-122).  With a GradReducer
        the gradients live in flat arenas: one fill to zero them, the exchange in place, one multiply to clip (1/G folded in)."""
        red = self.grad_reducer
        fused = getattr(self.optimizer, "clip_and_step", None)     # cvc.optim.ClipAdam: norm + clip + Adam in three launches
        if red is None:
            # ClipAdam keys its device segment table on the gradients' addresses: keep them (zero in place) instead of fresh
            # tensors every step, which would rebuild and re-upload the table per step
            self.optimizer.zero_grad(set_to_none=set_to_none and fused is None)
            loss.backward()
            if fused is not None:
                fused(self.opts.grad_clip, 1.0, **({'skip': self._status} if self._status is not None else {}))
                return
            nn.utils.clip_grad_norm_(self.model.parameters(), self.opts.grad_clip)
        else:
            red.zero_grad()
            loss.backward()
            # the one exchange step of the path (+ the step's status word, so that every rank voids -- and later re-runs -- the
            # same steps: a void step's gradients have already been summed into every rank's arenas)
            red.finalize(average=False, status=self._status)
            if fused is not None:
                # (the arenas hold sums over ranks: 1 / G folded into the coefficient; the gradients are read here for the last
                # time, so the pass leaves them zero and the next step's zero_grad() has nothing to fill)
                fused(self.opts.grad_clip, 1.0 / red.world, zero_grad=True, **({'skip': self._status} if self._status is not None else {}))
                red.mark_zeroed()
                return
            red.clip_(self.opts.grad_clip, summed=True)
        self.optimizer.step()

    def _weights_changed(self):
        """The fused Adam kernel and HIP-graph replays update parameters without touching their version counters,
        which is what the captioner's cached decode binding watches: drop it explicitly."""
        inval = getattr(self.model, "invalidate_decode_cache", None)
        if inval is not None:
            inval()                               # (bumps the weights generation as well)
        else:
            from . import hip
            hip.bump_weights_generation()

    # ------------------------------------------------------------------ HIP-graph training step
    def _core_step(self, b):
        out = self._call(b)
        loss, lm, att2, cls, rec = self.loss_mix(out)
        self._backward_and_update(loss, set_to_none=False)
        res = torch.stack([loss.detach().reshape(()), lm.detach().reshape(()), att2.detach().reshape(()),
                           cls.detach().reshape(()), rec.detach().reshape(())])
        st = self._status
        if st is not None:
            # deferred error words: a step some launch declared void contributes nothing to the loss sums and says so in a sixth
            # element; the word is cleared at the END of the step (whoever sets it before a step voids that step: tests do)
            void = st != 0
            res = torch.cat((torch.where(void, torch.zeros_like(res), res), void.to(res.dtype)))
            st.zero_()
        return res

    def train_step_graphed(self, batch):
        """Same step as train_step, captured once into a HIP graph and replayed: the per-step Python / launch overhead
        disappears.  Requirements: constant batch shapes (inputs are copied into static buffers), an optimizer built with
        capturable=True (build_optimizer(..., capturable=True)).  The gradient exchange of a multi-rank GradReducer is part of
        the captured step when it runs on RCCL (backend "nccl": the collectives are issued on the communicator's stream behind
        events of the capturing stream, which stream capture records as graph dependencies); other backends (gloo) cannot be
        captured.  Returns a static tensor [loss, lm, att2, cls, recon] (clone to keep)."""
        self._check_capturable()
        b = self._prepare(batch, True)
        if self._graph is None:
            static = {k: (v.clone() if isinstance(v, torch.Tensor) else ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v))
                      for k, v in b.items()}
            s = torch.cuda.Stream()
            if self.grad_reducer is not None:
                self.grad_reducer.bind_stream(s)                    # gradient accumulators on the stream of warm-up AND capture
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(3):                                  # warm-up on a side stream (allocator, lazy init, Adam state)
                    self._core_step(static)
            torch.cuda.current_stream().wait_stream(s)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s, capture_error_mode=self._capture_mode()):
                res = self._core_step(static)
            self._graph = (g, static, res)
        g, static, res = self._graph
        _copy_into(static, b)
        sync = getattr(self.optimizer, "sync_hyperparameters", None)
        if sync is not None:
            sync()                                # a scheduler may have moved the learning rates since the capture
        g.replay()
        self._weights_changed()
        return res

    def _check_capturable(self):
        """The gradient exchange inside a captured step must be a pure stream operation: the package's own RCCL communicator
        (GradReducer(comm=...), backend "rccl").  torch.distributed's collectives are refused -- gloo is host-side, and c10d's
        "nccl" group keeps a watchdog thread whose event queries race with the capture (round-4 finding: it killed about one
        capture in five; there is no way to fence it off short of waiting it out, which this code no longer does)."""
        red = self.grad_reducer
        if red is not None and red.exchange and red.backend != "rccl":
            raise RuntimeError(f"train_step_graphed: the gradient exchange on torch.distributed backend {red.backend!r} cannot be captured "
                               "into a HIP graph; build the GradReducer with comm=cvc.distributed.exchange_comm() (cvc.comm.RcclComm), "
                               "or use train_step")

    @staticmethod
    def _capture_mode() -> str:
        """"thread_local" when somebody's c10d "nccl" group is up in this process (its watchdog polls events from another thread:
        under "global" capture mode those queries fail and the watchdog takes the process down); nothing of that group is part of
        the captured step, so there is nothing to wait for."""
        import torch.distributed as dist
        up = dist.is_available() and dist.is_initialized() and dist.get_backend() == "nccl"
        return "thread_local" if up else "global"

    def graph_capable(self) -> bool:
        """train() replays captured steps when: --hip_graph is on, the optimizer keeps its step state on the device (ClipAdam or a
        capturable torch optimizer), and the gradient exchange (if any) is a stream operation."""
        if not self.shape_buckets or self.device.type != "cuda":
            return False
        # the model's whole step must consist of stream operations: the decode / localize / reconstruct path on pre-extracted
        # features is; the once-per-clip encoder is not (its persistent GRU reports barrier time-outs through a host read, widths
        # outside the HIP forms run in the library) -- raw-feature runs through the encoder train with eager steps
        cap = getattr(self.model, "step_capturable", None)
        if cap is None or not cap(deferred_errors=self._can_defer()):
            return False
        if getattr(self.optimizer, "clip_and_step", None) is None and not all(g.get("capturable", False) for g in self.optimizer.param_groups):
            return False
        red = self.grad_reducer
        return red is None or not red.exchange or red.backend == "rccl"

    def _can_defer(self) -> bool:
        """deferred error words need an optimizer pass that honours the status word (cvc.optim.ClipAdam) and, with an exchange, a
        transport that can carry the word inside the step (the package's RCCL communicator)"""
        red = self.grad_reducer
        return getattr(self.optimizer, "clip_and_step", None) is not None and self.device.type == "cuda" and \
            (red is None or not red.exchange or red.backend == "rccl")

    def _needs_deferral(self) -> bool:
        need = getattr(self.model, "reports_error_words", None)
        return bool(need is not None and need())

    def deferred_errors(self):
        """Context manager: error words of the model's persistent kernels go to the device's status word instead of being read by
        the host (train() uses it whenever it replays graphs over a model that has such kernels; bench.py for the same step)."""
        import contextlib
        from . import hip

        @contextlib.contextmanager
        def cm():
            if not (self._needs_deferral() and self._can_defer()):
                yield False
                return
            self._status = hip.step_status(self.device)
            self._status.zero_()
            hip.defer_errors(self._status)
            try:
                yield True
            finally:
                hip.defer_errors(None)
                self._status = None
        return cm()

    def rerun_void_steps(self, batches):
        """The steps of `batches` (prepared, on the device) were void: some persistent launch reported a barrier time-out, the
        optimizer pass left the parameters alone.  They are taken again here, eagerly, on the per-step forms (which have no barrier)
        with host-checked error words.  Every rank of a multi-GPU run calls this with the same number of batches (the status word
        travels with the gradient exchange).  Returns the five loss values of each re-run step."""
        from . import gru, hip
        hip.warn_once("trainer.void-steps", "a persistent recurrence kernel reported a barrier time-out inside a captured training step; "
                      "the step left the parameters untouched and is re-run on the per-step forms (reported once; counts in "
                      "Trainer.deferred_stats)")
        keep = (gru.PERSISTENT, gru.BWD_PERSISTENT, self._status)
        hip.defer_errors(None)
        gru.PERSISTENT = gru.BWD_PERSISTENT = False
        self._status = None
        out = []
        try:
            for b in batches:
                out.append(torch.stack([x.reshape(()) for x in self.train_step_prepared(b)]))
                self.deferred_stats["rerun_steps"] += 1
        finally:
            gru.PERSISTENT, gru.BWD_PERSISTENT, self._status = keep
            if self._status is not None:
                hip.defer_errors(self._status)
        return out

    def train_step_bucketed(self, b):
        """One optimisation step on a prepared batch, replayed from the HIP graph of its shape when there is one.  A shape's first
        occurrence runs eagerly -- that IS the training step, and it is the warm-up the capture needs (allocator, lazy workspaces);
        the very first steps of a run are eager whatever their shape (the reducer learns and compacts its arenas on step one, the
        optimizer builds its segment table on step two).  From its second occurrence on a shape is captured once (capturing runs
        nothing) and replayed.  Eager and replayed steps are the same launches on the same buffers: bit-identical results.
        Returns [loss, lm, att2, cls, recon] on the device (static for replayed steps: read or clone before the next step)."""
        key = _shape_key(b)
        if self._side is None:
            # ONE stream for eager steps, captures and replays; the reducer's gradient accumulators are re-created under it (a node
            # made under another stream would fork every backward pass -- and every captured graph -- once per parameter)
            self._side = torch.cuda.Stream(self.device)
            if self.grad_reducer is not None:
                self.grad_reducer.bind_stream(self._side)
        ent = self._graphs.get(key)
        if ent is None and (self._graph_broken or self._eager_steps < 2 or self._shape_seen.get(key, 0) < 1):
            cur = torch.cuda.current_stream()
            self._side.wait_stream(cur)
            with torch.cuda.stream(self._side):
                res = self._core_step(b)
            cur.wait_stream(self._side)
            res.record_stream(cur)
            for v in b.values():                     # the batch's tensors were used on the side stream: allocator ownership
                for t_ in (v.values() if isinstance(v, dict) else (v,)):
                    if isinstance(t_, torch.Tensor) and t_.is_cuda:
                        t_.record_stream(self._side)
            self._eager_steps += 1
            self._shape_seen[key] = self._shape_seen.get(key, 0) + 1
            self.graph_stats["eager"] += 1
            self._weights_changed()
            return res
        if ent is None:
            self._check_capturable()
            static = {k: (v.clone() if isinstance(v, torch.Tensor) else ({kk: vv.clone() for kk, vv in v.items()} if isinstance(v, dict) else v))
                      for k, v in b.items()}
            torch.cuda.current_stream().synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                # (every shape's graph allocates from its OWN private pool: graphs replay in data-dependent order, and PyTorch only
                # guarantees a shared pool for graphs replayed in capture order -- a later graph's intermediates could alias an
                # earlier graph's static result.  288 GB of HBM pay for one live set per shape, at most 16 shapes)
                with torch.cuda.graph(g, stream=self._side, capture_error_mode=self._capture_mode()):
                    res = self._core_step(static)
            except Exception as ex:      # noqa: BLE001 -- a model whose step cannot be captured (a host read inside the forward, a
                # library fallback that allocates) trains eagerly from here on; said once, loudly
                from . import hip
                hip.warn_once("trainer.capture", f"the training step could not be captured into a HIP graph ({type(ex).__name__}: "
                              f"{str(ex)[:300]}); training continues with eager steps")
                self._graph_broken = True
                torch.cuda.synchronize()
                if self.grad_reducer is not None:
                    self.grad_reducer.zero_grad()
                else:
                    self.optimizer.zero_grad(set_to_none=False)
                return self.train_step_bucketed(b)
            ent = self._graphs[key] = (g, static, res)
            self.graph_stats["captured"] += 1
        g, static, res = ent
        _copy_into(static, b)
        sync = getattr(self.optimizer, "sync_hyperparameters", None)
        if sync is not None:
            sync()
        g.replay()
        self.graph_stats["replayed"] += 1
        self._weights_changed()
        return res

    # ------------------------------------------------------------------ epoch loops
    def train(self, epoch, tb_logger=None):
        meters = {k: AverageMeter() for k in ("batch", "data", "lm", "attn", "cls", "recon")}
        self.model.train()
        end = time.time()
        n_steps = len(self.train_loader) - 1                 # the reference drops the last batch (:55)
        graphed = self.graph_capable()
        self._active_buckets = self.shape_buckets if graphed else 0       # (read by _prepare, which the prefetcher calls)
        # batch k+1 is staged into HBM on a side stream while step k runs
        batches = DevicePrefetcher(self.train_loader, lambda raw: self._prepare(raw, True), self.device, limit=n_steps)
        if self.opts.att_model != 'cyclical':
            raise ValueError('Unknown att_model: {}'.format(self.opts.att_model))
        n = self.opts.batch_size * self.opts.seq_per_img
        acc, pending, last = None, 0, None           # device-side sums of [loss, lm, att2, cls, recon (, void)] since the last read-back
        kept = []                                    # deferred error words: the interval's prepared batches (a void step is re-run)
        KEEP_MAX = 16                                # ... at most this many are held (377 MB each at config-2 size), then a read-back
        void_flags = None

        def flush():
            """ONE host read per display interval (the reference reads four scalars per step, trainer.py:124-135): the meters get
            the interval's means with the interval's weight, `.val` the latest step's values.  Deferred error words: the same read
            tells which steps of the interval were void; they are re-run now and their losses join the interval's sums."""
            nonlocal acc, pending
            if acc is None:
                return
            if deferred:
                sums, cur, flags = torch.cat([acc, last, void_flags[:pending]]).split([6, 6, pending])
                sums, cur, flags = sums.tolist(), cur.tolist(), flags.tolist()
                void = [i for i, f in enumerate(flags) if f != 0]
                if void:
                    self.deferred_stats["void_steps"] += len(void)
                    for r in self.rerun_void_steps([kept[i] for i in void]):
                        r = r.tolist()
                        sums[:5] = [a + b_ for a, b_ in zip(sums[:5], r)]
                        if void[-1] == pending - 1:
                            cur[:5] = r
                kept.clear()
            else:
                sums, cur = torch.stack([acc, last]).tolist()
            for name, i in (("lm", 1), ("attn", 2), ("cls", 3), ("recon", 4)):
                meters[name].update(sums[i] / pending, n * pending)
                meters[name].val = cur[i]
            acc, pending = None, 0

        step = -1
        import contextlib
        with (self.deferred_errors() if graphed else contextlib.nullcontext(False)) as deferred:
            if deferred:
                void_flags = torch.zeros(KEEP_MAX, device=self.device)
            for step, b in enumerate(batches):
                meters["data"].update(time.time() - end)
                if graphed:
                    res = self.train_step_bucketed(b)
                else:
                    res = torch.stack([x.reshape(()) for x in self.train_step_prepared(b)])
                acc = res.clone() if acc is None else acc.add_(res)
                if deferred:
                    void_flags[pending:pending + 1].copy_(res[5:6])
                    kept.append(b)
                last, pending = res, pending + 1
                if step % self.opts.disp_interval == 0:
                    flush()
                    meters["batch"].update((time.time() - end))
                    print('Epoch: [{0}][{1}/{2}]\tTime {b.val:.3f} ({b.avg:.3f})\tData {d.val:.3f} ({d.avg:.3f})\t'
                          'LM Loss {l.val:.4f} ({l.avg:.4f})\tAttn Loss {a.val:.4f} ({a.avg:.4f})\t'
                          'Cls Loss {c.val:.4f} ({c.avg:.4f})\tRecon Loss {r.val:.4f} ({r.avg:.4f})'.format(
                              epoch, step, n_steps, b=meters["batch"], d=meters["data"], l=meters["lm"], a=meters["attn"],
                              c=meters["cls"], r=meters["recon"]))
                else:
                    if deferred and pending >= KEEP_MAX:
                        flush()
                    meters["batch"].update(time.time() - end)      # (launch time of an asynchronous step; display steps include the wait)
                end = time.time()
            flush()
        if tb_logger:
            tb_logger.add_scalar('train/learning_rate', self.optimizer.param_groups[0]['lr'], epoch)
            tb_logger.add_scalar('train/lm_loss', meters["lm"].avg, epoch)
            tb_logger.add_scalar('train/attn_loss', meters["attn"].avg, epoch)
            tb_logger.add_scalar('train/cls_loss', meters["cls"].avg, epoch)
            tb_logger.add_scalar('train/lm_recon_loss', meters["recon"].avg, epoch)

    def _segment_timestamps(self):
        """{video: {segment: [start, end]}} of the validation segments: the reference reads them from the ANet-Entities
        annotation file `opts.grd_reference` (trainer.py:162) as ['annotations'][vid]['segments'][seg]['timestamps']; a
        dataset object may carry the same structure as `dataset.grd_reference` (the synthetic stand-in does)."""
        if getattr(self, "_timestamps", None) is None:
            src = getattr(self.dataset, "grd_reference", None)
            if src is None:
                path = getattr(self.opts, "grd_reference", None)
                if not path or not os.path.isfile(path):
                    raise FileNotFoundError(
                        f"eval needs the segment timestamps (opts.grd_reference = {path!r}, reference trainer.py:162): the "
                        "densecap file the ANETcaptions evaluator reads carries a 'timestamp' per predicted segment")
                with open(path) as f:
                    src = json.load(f)
            self._timestamps = src['annotations']
        return self._timestamps

    def eval(self, epoch, tb_logger=None):
        """Greedy (or beam) decode of the validation split -> predictions in the densecap JSON
        layout (trainer.py:254-286: {'sentence', 'timestamp': [start, end]} per segment) -> optional external scorer."""
        self.model.eval()
        predictions = defaultdict(list)
        grd_output = defaultdict(dict)
        o = self.opts
        ds = self.dataset
        timestamps = self._segment_timestamps()
        with torch.no_grad():
            for b in DevicePrefetcher(self.val_loader, lambda raw: self._prepare(raw, False), self.device):
                seq, att2_weights, _ = self._call(b, True)
                if getattr(o, "eval_obj_grounding", False):
                    assert o.beam_size == 1, 'only support beam_size is 1'
                    self._collect_grounding(b, seq, att2_weights, grd_output)
                sents = utils.decode_sequence(ds.itow, getattr(ds, "itod", None), getattr(ds, "ltow", None),
                                              getattr(ds, "itoc", None), getattr(ds, "wtod", None), seq.data, o.vocab_size, o)
                for k, sent in enumerate(sents):
                    vid_idx, entry = densecap_entry(sent, b["seg_id"][k], timestamps)
                    predictions[vid_idx].append(entry)
        # every rank decoded its shard of the clips; merge host-side, rank 0 writes the files and scores
        predictions, grd_output = gather_eval_outputs(predictions, grd_output)
        lang_stats = {}
        self.submission_file = self.attn_file = None
        self.predictions = predictions
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_rank() != 0:
            return lang_stats
        if getattr(o, "language_eval", False) and o.val_split != 'hidden_test':
            self.submission_file = write_densecap_json(predictions, o)
        if getattr(o, "eval_obj_grounding", False):
            self.attn_file = write_grounding_json(grd_output, o)
        if self.scorer is not None:
            lang_stats = self.scorer(predictions, grd_output, o) or {}
            if tb_logger:
                for k, v in lang_stats.items():
                    tb_logger.add_scalar('eval/' + k, v, epoch)
        self.predictions = predictions
        return lang_stats

    def _collect_grounding(self, b, seq, att2_weights, grd_output):
        """Per generated word, the most attended proposal of every sampled frame (trainer.py:217-248)."""
        o = self.opts
        B = seq.size(0)
        ppls = b["ppls"]
        att2_ind = torch.max(att2_weights.view(B, att2_weights.size(1), o.num_sampled_frm, o.num_prop_per_frm), dim=-1)[1]
        boxes = torch.gather(ppls.view(-1, o.num_sampled_frm, o.num_prop_per_frm, 7).permute(0, 2, 1, 3).contiguous(), 1,
                             att2_ind.unsqueeze(-1).expand(B, att2_ind.size(1), o.num_sampled_frm, ppls.size(-1)))
        lemma_det = {o.wtol[k]: i for k, i in o.wtod.items() if k in o.wtol}
        words = seq.tolist()                                   # one device -> host copy each, not one per word
        boxes_l = boxes[..., :4].tolist()
        for i in range(B):
            vid_id, seg_idx = b["seg_id"][i].split('_segment_')
            res = {'clss': [], 'idx_in_sent': [], 'bbox_for_all_frames': []}
            for j, w in enumerate(words[i]):
                if w == 0:
                    break
                lemma = o.wtol[o.itow[str(w)]]
                if lemma in lemma_det:
                    res['bbox_for_all_frames'].append(boxes_l[i][j])
                    res['clss'].append(o.itod[lemma_det[lemma]])
                    res['idx_in_sent'].append(j)
            grd_output[vid_id][str(int(seg_idx))] = res


def densecap_entry(sentence, seg_id, timestamps):
    """One element of predictions[video] as the reference builds it (trainer.py:253-261): the sentence plus the segment's
    [start, end] from the annotation file, rounded to 2 decimals -- what the ANETcaptions evaluator matches on.
    `segment` (the segment index as a string) is an extra key for the multi-rank merge and debugging."""
    vid_idx, seg_idx = seg_id.split('_segment_')
    seg_idx = str(int(seg_idx))
    stamps = timestamps[vid_idx]['segments'][seg_idx]['timestamps']
    return vid_idx, {'sentence': sentence, 'timestamp': [round(ts, 2) for ts in stamps], 'segment': seg_idx}


def _results_path(o, stem):
    d = getattr(o, "results_dir", "results")
    os.makedirs(d, exist_ok=True)
    return os.path.join(d, stem + '-' + o.val_split + '-' + o.id + '.json')


def write_densecap_json(predictions, o):
    """The ActivityNet dense-captioning submission file the external scorer reads (trainer.py:279-286)."""
    path = _results_path(o, 'densecap')
    with open(path, 'w') as f:
        json.dump({'version': 'VERSION 1.0', 'results': predictions,
                   'external_data': {'used': 'true', 'details': 'Visual Genome for Faster R-CNN pre-training'}}, f)
    return path


def write_grounding_json(grd_output, o):
    """Per-word grounding boxes of the generated sentences for the ANet-Entities scorer (trainer.py:318-329)."""
    path = _results_path(o, 'attn-gen-sent-results')
    with open(path, 'w') as f:
        json.dump({'results': grd_output, 'eval_mode': 'gen',
                   'external_data': {'used': True,
                                     'details': 'Object detector pre-trained on Visual Genome on object detection task.'}}, f)
    return path


CLIP_ADAM = os.environ.get("CVC_CLIP_ADAM", "1") != "0"      # False: torch.optim.Adam (fused) + the library's clip (A/B)


def build_optimizer(model, opt, capturable: bool = False):
    """One param group per tensor; 0.1x LR for ctx2pool_grd / vis_embed (reference main.py:171-191).
    capturable=True keeps Adam's step counters on the device (needed by Trainer.train_step_graphed)."""
    params = []
    for key, value in dict(model.named_parameters()).items():
        if not value.requires_grad:
            continue
        lr = opt.learning_rate * (0.1 if ('ctx2pool_grd' in key or 'vis_embed' in key) else 1.0)
        params.append({'params': [value], 'lr': lr, 'weight_decay': opt.weight_decay, 'betas': (opt.optim_alpha, opt.optim_beta)})
    if opt.optim == 'sgd':
        return torch.optim.SGD([{k: v for k, v in p.items() if k != 'betas'} for p in params], lr=opt.learning_rate, momentum=0.9)
    if opt.optim == 'adam':
        # fused = one pass over (p, g, m, v) per tensor instead of the foreach path's ~9 passes (Adam was 8 % of the step)
        fused = all(p['params'][0].is_cuda for p in params)
        if fused and CLIP_ADAM:
            from .optim import ClipAdam                      # clip_grad_norm_ + step in one pass over p / g / m / v (csrc/optim.hip)
            return ClipAdam(params)
        return torch.optim.Adam(params, capturable=capturable, fused=fused)
    if opt.optim == 'adamax':
        return torch.optim.Adamax(params)
    raise ValueError('Unknown optimizer: {}'.format(opt.optim))
