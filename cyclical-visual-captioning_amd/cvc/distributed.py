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


def init_from_env(backend: str = "rccl") -> Tuple[int, int, int]:
    """(rank, world, local_rank) from torchrun's environment; initialises the process group.

    backend "rccl" (the default on GPUs): torch.distributed on "gloo" as the CONTROL plane only (rendezvous, the communicator's
    unique id, host-side barriers, object broadcasts) -- the gradient exchange itself runs on the package's own RCCL communicator
    (cvc.comm.RcclComm, `exchange_comm()` below), which has no watchdog thread and can be captured into the step's HIP graph.
    "nccl": c10d's RCCL process group for control and data (eager steps only).  "gloo": CPU tests."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            if backend == "rccl":
                torch.cuda.set_device(local_rank)
            dist.init_process_group("gloo")
    return rank, world, local_rank


_EXCHANGE_COMM = None


def exchange_comm(create: bool = True):
    """The process's RCCL communicator for the gradient exchange (cvc.comm.RcclComm), built once over the initialised
    torch.distributed group (its unique id travels on the control plane); None without a group or without a GPU."""
    global _EXCHANGE_COMM
    if _EXCHANGE_COMM is None and create and dist.is_available() and dist.is_initialized() and torch.cuda.is_available():
        from .comm import RcclComm
        _EXCHANGE_COMM = RcclComm.from_process_group()
    return _EXCHANGE_COMM


def destroy_exchange_comm():
    global _EXCHANGE_COMM
    if _EXCHANGE_COMM is not None:
        _EXCHANGE_COMM.destroy()
        _EXCHANGE_COMM = None


def control_all_reduce(values, op: str = "max"):
    """all-reduce of a few host numbers over the control plane (CPU tensor on gloo, device tensor on c10d-"nccl") -> list"""
    if not (dist.is_available() and dist.is_initialized()):
        return list(values)
    on_gpu = dist.get_backend() == "nccl"
    t = torch.tensor(list(values), dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()) if on_gpu else "cpu")
    dist.all_reduce(t, op={"max": dist.ReduceOp.MAX, "min": dist.ReduceOp.MIN, "sum": dist.ReduceOp.SUM}[op])
    return t.tolist()


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
    """Flat gradient arenas + hook-driven asynchronous exchange (replaces the reference's nn.DataParallel, main.py:169).

    * Every bucket is ONE flat fp32 tensor and every `param.grad` of the bucket is a view into it, for the whole run: autograd
      accumulates straight into the arena, so the exchange needs no `cat`, no copy-back and no per-tensor kernels;
      `zero_grad()` is one fill per bucket, `clip_()` one norm + one multiply per bucket with the 1/G of the averaging folded
      into the clip coefficient (trainer.py:120-121 semantics: global norm over the AVERAGED gradients).
    * Buckets follow the order in which backward completes gradients: the vocabulary head is complete
      before BPTT of the decode loop starts and is exchanged under it; everything that is touched at every time step (LSTM
      cells, h2attn, alpha_net, embedding) completes at t = 0.  Each LSTM weight matrix is its own bucket (its deferred
      dW GEMM finishes separately), biases ride with their cell's weight_hh.
    * On RCCL a bucket goes as in-place reduce_scatter + all_gather over the xGMI mesh, two calls issued back to back the
      moment the bucket is complete.  Transport `comm=` (cvc.comm.RcclComm, the package's own communicator -- the default on
      GPUs): the pair is enqueued on a dedicated exchange stream behind an event of the step's stream, and finalize() makes the
      step's stream wait for it; no host object, no watchdog thread, capturable into the step's HIP graph.  Transport c10d
      (`comm=None` with an initialised group): "nccl" issues the same pair through torch.distributed (eager steps only), other
      backends (gloo in the CPU tests) use one all_reduce.
    * Parameters that never receive a gradient are found on the first step (every rank must see the same
      set: checked) and keep `.grad = None` from then on, as in the reference -- optimizers skip them (no weight decay, no Adam
      state); their arena slots stay zero, so message sizes never depend on None-ness.
    Expected exposed time at BASELINE config 4 (8 x MI355X, 486 MB of gradients, 7 xGMI links per GPU): the 445 MB that
    complete at t = 0 cost 2 x 7/8 x 445 MB per GPU over 7 links = 0.7 ms at the 153 GB/s link peak, ~2 ms at a third of it,
    against a ~29 ms step; the head's 41 MB overlap the decode loop's BPTT.
    """

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], bucket_mb: float = 0.0, group=None,
                 overlap: bool = True, world: Optional[int] = None, always_exchange: bool = False, comm=None):
        """always_exchange: issue the collectives even in a one-rank group (tests: the RCCL path on a single GPU).
        bucket_mb > 0 additionally splits the 'rest' bucket into pieces of at most that size (tests use tiny buckets).
        comm: a cvc.comm.RcclComm -- the exchange then runs on it (backend "rccl") instead of on torch.distributed."""
        self.group = group
        self.comm = comm
        if comm is not None:
            self.world = comm.world
            self.backend = "rccl"
        else:
            self.world = world if world is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
            self.backend = dist.get_backend(group) if dist.is_initialized() else "none"
        self._comm_stream = None                              # the exchange stream of the `comm` transport (made on first use)
        self._status_f = None                                 # exchange buffer of the step's status word (finalize(status=...))
        self._comm_pending = False
        seen, params = set(), []
        for name, p in named_params:
            if p.requires_grad and id(p) not in seen:       # shared LSTM cells are listed once
                seen.add(id(p))
                params.append((name, p))
        # ---- bucket plan, in backward-completion order
        head, rest, cells = [], [], {}
        for name, p in params:
            if "lstm" in name:
                cell = name.rsplit(".", 1)[0]
                kind = name.rsplit(".", 1)[1]
                key = (cell, "ih") if kind == "weight_ih" else (cell, "hh")
                cells.setdefault(key, []).append((name, p))
            elif name.startswith("logit") or ".logit." in name:
                head.append((name, p))
            else:
                rest.append((name, p))
        groups: List[List[Tuple[str, torch.nn.Parameter]]] = []
        if head:
            groups.append(head)
        if rest:
            if bucket_mb > 0:
                cap, cur, size = int(bucket_mb * (1 << 20) / 4), [], 0
                for item in reversed(rest):
                    if cur and size + item[1].numel() > cap:
                        groups.append(cur)
                        cur, size = [], 0
                    cur.append(item)
                    size += item[1].numel()
                if cur:
                    groups.append(cur)
            else:
                groups.append(list(reversed(rest)))
        groups += [cells[k] for k in sorted(cells, reverse=True)]
        self.buckets = groups
        self.arenas: List[torch.Tensor] = []
        self._views: Dict[int, torch.Tensor] = {}
        align = 64 * max(self.world, 1)                       # floats: equal, 256-byte aligned shards for reduce_scatter
        for b in groups:
            n = sum((p.numel() + 63) // 64 * 64 for _, p in b)
            n = (n + align - 1) // align * align
            p0 = b[0][1]
            arena = torch.zeros(n, device=p0.device, dtype=p0.dtype)
            off = 0
            for _, p in b:
                v = arena[off:off + p.numel()].view_as(p)
                self._views[id(p)] = v
                p.grad = v
                off += (p.numel() + 63) // 64 * 64
            self.arenas.append(arena)
        self._bucket_of: Dict[int, int] = {id(p): i for i, b in enumerate(groups) for _, p in b}
        self._ready = [0] * len(groups)
        self._launched = [False] * len(groups)
        self._expected: Optional[List[int]] = None           # gradient arrivals per bucket, learned on the first step
        self._seen_first: List[set] = [set() for _ in groups]
        self._seen_late: List[set] = [set() for _ in groups]   # first step: weights whose gradient arrived through the deferred-dW flush
        self._live: Optional[List[frozenset]] = None          # per bucket: ids whose hook must fire before the bucket may leave
        self._arrived: List[set] = [set() for _ in groups]
        self._names: Dict[int, str] = {id(p): n for b in groups for n, p in b}
        self._work: List = []
        self.exchange = self.world > 1 or (always_exchange and (comm is not None or dist.is_initialized()))
        self.overlap = overlap and self.exchange
        # hooks also without an exchange: the first step learns which parameters ever receive a gradient (see finalize)
        self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in groups for _, p in b]
        self._dead: set = set()                               # ids of parameters that never receive a gradient: .grad stays None
        self._learned = False
        self._late_after_launch: List[int] = []
        from . import functional as _F
        _F.LATE_GRAD_LISTENERS.append(self._on_late_grad)
        _F.GRAD_SINKS.append(self)
        self._late_listener_owner = _F
        self._written: set = set()                            # ids whose arena view was handed out for an in-place write this step
        self._final: set = set()                              # ... whose producer declared the write complete (written(final=True))
        self._hooked: set = set()                             # ids whose post-accumulate-grad hook has fired this step
        self._done_how: List[Optional[str]] = [None] * len(groups)     # how each bucket became complete this step
        self.track_ready = False                              # True: record an event when a bucket becomes complete (bench.py)
        self._step_stream = None
        self.ready_events: Dict[int, "torch.cuda.Event"] = {}

    def bind_stream(self, stream) -> None:
        """Re-create the parameters' gradient accumulators under `stream`.  autograd runs an AccumulateGrad node on the stream that
        was current when the node was CREATED, and a registered post-accumulate hook keeps the node alive for the whole run: nodes
        made when this reducer was built (typically under the default stream) make every backward pass fork to that stream and
        back for each accumulated gradient -- in a captured step that is a graph with one side branch per parameter.  The trainer
        calls this once with the stream its steps run on (eager, capture and replay alike), before its first step."""
        if not self._hooks or not self.arenas or not self.arenas[0].is_cuda:
            return
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        import gc
        gc.collect()                                          # the old accumulators die with their last reference (the hooks)
        with torch.cuda.stream(stream):
            self._hooks = [p.register_post_accumulate_grad_hook(self._on_grad) for b in self.buckets for _, p in b]
        self._bound_stream = stream

    # ------------------------------------------------------------------ gradient bookkeeping
    def mark_zeroed(self):
        """The optimizer step left every gradient it consumed zero (cvc.optim.ClipAdam.clip_and_step(zero_grad=True)): the next
        zero_grad() skips its fills unless something writes a gradient in between."""
        self._clean = True

    def zero_grad(self):
        """One fill per bucket (none when the optimizer step already zeroed what it read, see mark_zeroed); re-attaches a view if
        something replaced a .grad (optimizer.zero_grad(set_to_none=True))."""
        if not getattr(self, "_clean", False):
            for a in self.arenas:
                a.zero_()
        self._clean = False
        self.ready_events = {}
        self._step_stream = torch.cuda.current_stream() if self.arenas and self.arenas[0].is_cuda else None
        for b in self.buckets:
            for _, p in b:
                if id(p) in self._dead:                       # never receives a gradient: None, as in the reference (optimizers skip
                    continue                                  # it: no weight decay, no Adam state -- trainer.py:118-122, opts.py:109)
                v = self._views[id(p)]
                if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                    p.grad = v
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._arrived = [set() for _ in self.buckets]
        self._written, self._final, self._hooked = set(), set(), set()
        self._done_how = [None] * len(self.buckets)

    # ---- gradient sink protocol (cvc.functional.GRAD_SINKS): the dense weight-gradient products write straight into the arena
    def claim(self, p):
        i = self._bucket_of.get(id(p))
        if i is None or id(p) in self._dead or id(p) in self._written or self._launched[i]:
            return None
        if id(p) in self._hooked or id(p) in self._arrived[i]:
            # another producer's gradient already sits in the view (AccumulateGrad ran, or an earlier in-place write): an
            # overwriting product would drop it -- the caller accumulates instead (compute(None) + grad.add_)
            return None
        v = self._views[id(p)]
        if p.grad is None or p.grad.data_ptr() != v.data_ptr():
            return None
        self._written.add(id(p))
        return v

    def written(self, p, final: bool = False):
        """the producer has enqueued its in-place write of p's gradient.  Recorded as an arrival (the first step learns from it which
        parameters are live).  final=True is the producer's word that this write is p's COMPLETE gradient of this backward pass (the
        training loops' one flush per step): the bucket may then leave at once -- its exchange is enqueued behind the product that
        just finished while the remaining weight-gradient products run.  Without it the bucket leaves from the
        post-accumulate-grad hook, which autograd fires for p once ALL of its producers have returned, or at finalize()."""
        i = self._bucket_of[id(p)]
        self._clean = False
        if id(p) not in self._arrived[i]:
            self._arrived[i].add(id(p))
            self._ready[i] += 1
        if final:
            self._final.add(id(p))
        if not self._learned:
            self._seen_first[i].add(id(p))
        elif final:
            self._maybe_complete(i, "in-place write")

    def _maybe_complete(self, i: int, how: str):
        """bucket i holds every gradient it will get this step: note when (track_ready) and let it leave (overlap)"""
        if self._done_how[i] is not None or not (self._arrived[i] >= self._live[i]) or self._seen_late[i]:
            return
        # parameters announced by a hook may still be written by a later producer unless every one of them is final or the hooks of
        # all of them have fired (a hook fires after the parameter's last producer)
        if not all((q in self._final) or (q in self._hooked) for q in self._live[i]):
            return
        self._done_how[i] = how
        if self.track_ready:
            # on the stream the step runs on (noted by zero_grad): a hook may run under another current stream -- autograd gives an
            # AccumulateGrad node the stream its leaf was first used on, e.g. the warm-up stream of a graph capture
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(self._step_stream if self._step_stream is not None else torch.cuda.current_stream())
            self.ready_events[i] = ev
        if self.overlap and not self._launched[i]:
            self._launch(i)

    def _revive(self, p):
        """A parameter that was marked dead (no gradient on the first step) receives one after all: its arena slot would never
        have been exchanged or cleared -- refuse loudly instead of training on a gradient the other ranks do not see."""
        raise RuntimeError(f"GradReducer: parameter {self._names.get(id(p), '?')} received no gradient on the first step (and was "
                           "dropped from the exchange, as the reference's optimizers skip None gradients) but receives one now: "
                           "the set of trained parameters must not change between steps")

    def _on_late_grad(self, w):
        """cvc.functional's deferred-dW flush wrote into w.grad at the end of backward (a use of the weight got no gradient): no
        hook fired for it.  Harmless unless the bucket's exchange was already launched -- then the late add races with / is
        missing from the collective, and finalize() refuses to continue."""
        i = self._bucket_of.get(id(w))
        if i is None:
            return
        self._clean = False
        if id(w) in self._dead:
            self._revive(w)
        v = self._views[id(w)]
        if w.grad is not None and w.grad.data_ptr() != v.data_ptr():   # the flush installed its own tensor (w.grad was None): into the arena
            v.copy_(w.grad)
            w.grad = v
        if not self._learned:
            self._seen_late[i].add(id(w))                   # live, but no hook will ever announce it: not part of _expected
        if self.exchange and self._launched[i]:
            self._late_after_launch.append((i, self._names.get(id(w), "?"), "deferred dW flush at the end of backward"))

    def _on_grad(self, p):
        i = self._bucket_of[id(p)]
        v = self._views[id(p)]
        self._clean = False
        if id(p) in self._dead:
            self._revive(p)
        if p.grad is not v and p.grad.data_ptr() != v.data_ptr():      # autograd installed its own tensor: move it into the arena
            v.copy_(p.grad)
            p.grad = v
        if id(p) not in self._arrived[i]:
            self._arrived[i].add(id(p))
            self._ready[i] += 1
        self._hooked.add(id(p))
        if not self._learned:
            self._seen_first[i].add(id(p))
        else:
            # every parameter of the bucket that was live on the first step has arrived (identities, not counts); a bucket that
            # also holds late-flushed weights leaves at finalize()
            self._maybe_complete(i, "hook")

    def _launch(self, i: int):
        self._launched[i] = True
        if not self.exchange:
            return
        a = self.arenas[i]
        if self.comm is not None and not a.is_cuda:
            # the CPU tests' stand-in transport (RcclComm(host_buffers=True) over tests/stub_rccl): the same cvc_allreduce_grads
            # call on a host arena, synchronous -- no streams to fork or join
            self.comm.all_reduce_(a, None)
        elif self.comm is not None:
            # own communicator: the pair goes on the exchange stream, behind everything the step's stream (and the hook's own
            # stream, see below) has enqueued so far; plain HIP events order it -- eagerly or as edges of the captured graph
            cur = torch.cuda.current_stream()
            st = self._step_stream if self._step_stream is not None else cur
            if st != cur:
                st.wait_stream(cur)
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream(device=a.device)
            self._comm_stream.wait_stream(st)
            self.comm.all_reduce_(a, self._comm_stream)
            self._comm_pending = True
        elif self.backend == "nccl":
            # Issued under the stream the step runs on, whatever stream the caller is under: a post-accumulate-grad hook runs under
            # the stream autograd gave its AccumulateGrad node (the warm-up stream of a graph capture, for one), and c10d orders the
            # collective behind -- and decides "is a capture going on?" from -- the CURRENT stream.  Under the step's stream the
            # exchange is ordered behind the products that wrote the arena, and c10d sees the capture (it then keeps the work
            # away from its watchdog thread, whose event queries are illegal on events recorded in a capturing stream).
            cur = torch.cuda.current_stream()
            st = self._step_stream if self._step_stream is not None else cur
            if st != cur:
                st.wait_stream(cur)               # what the hook's own stream did to the gradient (an accumulate, a copy into the arena)
            with torch.cuda.stream(st):
                n = a.numel() // self.world
                r = dist.get_rank(self.group)
                shard = a[r * n:(r + 1) * n]
                w1 = dist.reduce_scatter_tensor(shard, a, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                w2 = dist.all_gather_into_tensor(a, shard, group=self.group, async_op=True)
            self._work += [w1, w2]
        else:
            self._work.append(dist.all_reduce(a, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def _exchange_status(self, status: torch.Tensor):
        """The step's status word (cvc.hip.step_status: non-zero = a launch of this rank's step reported invalid outputs) summed over
        the ranks, so that EVERY rank voids the step -- the void rank's gradients are already part of every rank's arenas -- and
        every rank re-runs it later (cvc.trainer.Trainer.train).  One more in-place exchange on the exchange stream, behind the
        buckets; a 64 x world float buffer keeps it on the same reduce-scatter + all-gather branch as the arenas."""
        if self._status_f is None:
            self._status_f = torch.zeros(64 * max(self.world, 1), device=status.device, dtype=torch.float32)
        st = self._step_stream if self._step_stream is not None else torch.cuda.current_stream()
        if self._comm_stream is None:
            self._comm_stream = torch.cuda.Stream(device=status.device)
        self._comm_stream.wait_stream(st)
        with torch.cuda.stream(self._comm_stream):
            self._status_f[0:1].copy_(status)
            self.comm.all_reduce_(self._status_f, self._comm_stream)
            status.copy_(self._status_f[0:1] != 0)
        self._comm_pending = True

    def finalize(self, average: bool = True, status: Optional[torch.Tensor] = None):
        """Call after backward(): exchanges what the hooks did not, waits.  average=True leaves every gradient divided by
        the world size; average=False leaves SUMS for clip_() to fold the 1/G into its single multiply.  status: the step's status
        word (deferred error words); with an exchange on the package's communicator it is combined over the ranks as well."""
        if self._late_after_launch:
            late, self._late_after_launch = sorted(set(self._late_after_launch)), []
            raise RuntimeError(f"GradReducer: a weight gradient was written after its bucket's exchange had been launched "
                               f"(bucket, parameter, how): {late}; run with overlap=False for this graph")
        if self.exchange:
            if self._expected is None:
                # first step: every rank must expect the same arrivals per bucket (same model, same graph) -- checked once
                # (identities, not only counts: a hash of the names of the live parameters of every bucket)
                import zlib
                sig = [zlib.crc32("\n".join(sorted(self._names[q] for q in (s_ | l_))).encode()) + (len(s_) << 32)
                       for s_, l_ in zip(self._seen_first, self._seen_late)]
                if self.world > 1 and dist.is_initialized():
                    # (over the control plane: a CPU tensor unless the group is c10d-"nccl")
                    on_gpu = dist.get_backend(self.group) == "nccl"
                    counts = torch.tensor(sig, dtype=torch.int64, device=self.arenas[0].device if on_gpu else "cpu")
                    lo, hi = counts.clone(), counts.clone()
                    dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.group)
                    dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.group)
                    if not torch.equal(lo, hi):
                        raise RuntimeError("GradReducer: ranks disagree on which parameters receive gradients: "
                                           f"{lo.tolist()} vs {hi.tolist()}")
                self._expected = [len(s_) for s_ in self._seen_first]
            for i in range(len(self.buckets)):
                if not self._launched[i]:
                    self._launch(i)
            for w in self._work:
                w.wait()
            self._work.clear()
            if status is not None and self.world > 1:
                if self.comm is None or not status.is_cuda:
                    raise RuntimeError("GradReducer.finalize(status=...): the status word travels on the package's RCCL communicator only")
                self._exchange_status(status)
            if self._comm_pending:
                # the step's stream continues (clip + Adam read the arenas) behind the exchange stream
                st = self._step_stream if self._step_stream is not None else torch.cuda.current_stream()
                st.wait_stream(self._comm_stream)
                cur = torch.cuda.current_stream()
                if cur != st:
                    cur.wait_stream(self._comm_stream)
                self._comm_pending = False
            if average:
                for a in self.arenas:
                    a.mul_(1.0 / self.world)
        if not self._learned:
            # first step done: parameters that received no gradient keep .grad = None from now on (the reference's optimizers
            # skip them -- with weight_decay > 0 an all-zero gradient would still decay them and allocate Adam state); their
            # arena slots stay zero, so the exchanged message sizes are unchanged.  Multi-rank: agreement was checked above.
            self._learned = True
            if self._expected is None:
                self._expected = [len(s) for s in self._seen_first]
            self._live = [frozenset(s_) for s_ in self._seen_first]
            for i, b in enumerate(self.buckets):
                for _, p in b:
                    if id(p) not in self._seen_first[i] and id(p) not in self._seen_late[i]:
                        self._dead.add(id(p))
                        p.grad = None
            self._compact()
        self.last_done_how = list(self._done_how)             # (how every bucket became complete in the step that just ended)
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self._arrived = [set() for _ in self.buckets]
        self._written, self._final, self._hooked = set(), set(), set()
        self._done_how = [None] * len(self.buckets)

    def _compact(self):
        """End of the first step: the never-used parameters are known (the same set on every rank: checked) -- their slots leave the
        arenas, so that no later exchange carries their zeros (66 MB of 581 MB at D = 2048: i2h_2, h2h_2, localied_fc, the
        reconstructor's attention).  Live gradients keep their values and their order; every .grad is re-pointed at its new view
        (the optimizer builds its segment table after this)."""
        align = 64 * max(self.world, 1)
        for i, b in enumerate(self.buckets):
            if not any(id(p) in self._dead for _, p in b):
                continue
            live = [(n_, p) for n_, p in b if id(p) not in self._dead]
            n = sum((p.numel() + 63) // 64 * 64 for _, p in live)
            n = max(align, (n + align - 1) // align * align)
            old = self.arenas[i]
            arena = torch.zeros(n, device=old.device, dtype=old.dtype)
            off = 0
            for _, p in live:
                v = arena[off:off + p.numel()].view_as(p)
                v.copy_(self._views[id(p)])
                self._views[id(p)] = v
                p.grad = v
                off += (p.numel() + 63) // 64 * 64
            self.arenas[i] = arena

    def clip_(self, max_norm: float, summed: bool = True) -> torch.Tensor:
        """clip_grad_norm_(parameters, max_norm) over the arenas (trainer.py:120-121): global L2 norm of the averaged
        gradients, one multiply per bucket.  summed=True: the arenas hold sums over ranks (finalize(average=False))."""
        inv = 1.0 / self.world if summed else 1.0
        total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(a) for a in self.arenas])) * inv
        coef = torch.clamp(max_norm / (total + 1e-6), max=1.0) * inv
        for a in self.arenas:
            a.mul_(coef)
        return total

    def remove_hooks(self):
        for h in self._hooks:
            h.remove()
        self._hooks.clear()
        lst = self._late_listener_owner.LATE_GRAD_LISTENERS
        if self._on_late_grad in lst:
            lst.remove(self._on_late_grad)
        if self in self._late_listener_owner.GRAD_SINKS:
            self._late_listener_owner.GRAD_SINKS.remove(self)
