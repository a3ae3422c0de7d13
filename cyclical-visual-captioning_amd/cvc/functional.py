"""Autograd-aware ops of the caption-decode hot path, each backed by the HIP kernels in
libcvc_hip.so (cvc.hip).  Forward passes, the fused / recomputing backward passes, the LSTM cells' backward-data
product dX = dY W (csrc/gemm_nn.hip) and the dense products of the backward (dW = dY^T X batched over all T steps,
the M = B*T dX of the vocabulary head: cvc.hip.tile_mm on csrc/gemm_tile.hip) are hand-written kernels; no library
GEMM is on the training path (the grounder's backward, which the reference's objective never reaches, is the exception).

No CPU path: every op raises if handed CPU tensors (cvc.hip._dev).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch

from . import hip

Tensor = torch.Tensor


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


def _rows(t: Tensor) -> Tensor:
    """A [M, k] GEMM input as the kernels take it: unit inner stride, rows 16-byte aligned (any row stride) -- a step's slice
    of a [B, T, k] tensor goes in as it is instead of through a copy."""
    if t.dim() == 2 and t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0:
        return t
    return t.contiguous()


# ------------------------------------------------------------------------------- deferred weight gradients
class _WeightGradBatcher:
    """A weight used at every time step (LSTM cells, h2attn: 2 x T uses per training step) would
    otherwise get T skinny dW = dY_t^T X_t GEMMs (contraction over 64 rows each) plus T full-size
    accumulations.  Instead each backward call stashes (dY_t, X_t); the LAST call of the backward pass
    (use counter reaches zero) does ONE GEMM over all stashed steps (contraction over T x B rows) and
    returns it as its own gradient while the other calls return None -- autograd's sum is unchanged.
    If some uses never receive a gradient the leftovers are flushed into .grad by an end-of-backward
    callback."""

    def __init__(self):
        self.uses = {}        # key -> outstanding forward uses
        self.stash = {}       # key -> list of tuples of tensors
        self.owner = {}       # key -> (weight tensors..., flush function)
        self._callback_armed = False

    def note_use(self, key):
        self.uses[key] = self.uses.get(key, 0) + 1

    def add(self, key, item, owner, flush_fn):
        """Returns the flushed gradients when this was the last outstanding use, else None."""
        self.stash.setdefault(key, []).append(item)
        self.owner[key] = (owner, flush_fn)
        self.uses[key] = self.uses.get(key, 1) - 1
        if not self._callback_armed:
            self._callback_armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
        if self.uses[key] <= 0:
            items = self.stash.pop(key)
            self.uses.pop(key, None)
            self.owner.pop(key, None)
            return flush_fn(items)
        return None

    def _end_of_backward(self):
        self._callback_armed = False
        for key in list(self.stash):
            items = self.stash.pop(key)
            owner, flush_fn = self.owner.pop(key)
            self.uses.pop(key, None)
            grads = flush_fn(items)
            for w, g in zip(owner, grads):
                if w is not None and g is not None and w.requires_grad:
                    if w.grad is None:
                        w.grad = g
                    else:
                        w.grad.add_(g.view_as(w.grad))       # in place: .grad may be a view into a gradient arena
                    for fn in LATE_GRAD_LISTENERS:           # (no post-accumulate hook fires for this write: tell whoever
                        fn(w)                                #  exchanges gradients, cvc.distributed.GradReducer)
        self.uses.clear()


LATE_GRAD_LISTENERS = []      # callables(weight): a leftover dW was added to weight.grad at the end of backward, outside AccumulateGrad
_BATCHER = _WeightGradBatcher()

# Gradient sinks: whoever owns the parameters' .grad storage for the whole run (cvc.distributed.GradReducer: flat arenas, zero-filled
# at the start of a step) can hand a producer the .grad buffer itself -- the dense weight-gradient products then write their result
# in place and return None to autograd: no AccumulateGrad add over the 486 MB of gradients, and the owner learns of the arrival at
# once (bucket exchange launched as the product finishes).  sink.claim(param) -> a zero-filled tensor shaped like param that may be
# OVERWRITTEN (at most once per step), or None; sink.written(param) after the producing kernel was enqueued.
GRAD_SINKS = []


def claim_grad(p):
    """-> (buffer to overwrite, its sink) or (None, None)"""
    if p is None or not p.requires_grad:
        return None, None
    for sink in GRAD_SINKS:
        v = sink.claim(p)
        if v is not None:
            return v, sink
    return None, None


def _into_grad(param, compute):
    """compute(out) -> tensor: run with out = the gradient owner's buffer for `param` when a sink hands it out (then autograd gets
    None for this parameter), else with out = None and the result goes through autograd.  Only for parameters whose whole gradient
    of a backward pass comes from this ONE call."""
    buf, sink = claim_grad(param)
    if buf is None:
        return compute(None)
    compute(buf)
    sink.written(param)
    return None


def _mm_tn(d: Tensor, x: Tensor) -> Tensor:
    """dW = d^T x  ([S, N]^T [S, K] -> [N, K]): contraction over the S = T*B sample rows, on the tile GEMM."""
    return hip.tile_mm(d, x, a_kmajor=True, b_kmajor=True)


def _mm_nn(dy: Tensor, w: Tensor) -> Tensor:
    """dX = dy w  ([M, N] [N, K] -> [M, K]; w may be a column slice of a wider matrix).  Up to 64 rows: the weight-streaming
    backward-data kernel (csrc/gemm_nn.hip, weights read in place, dY repacked into its quad layout by one small launch);
    more rows, or shapes that kernel does not take: the tile GEMM."""
    M, N = dy.shape
    if M <= 64 and dy.stride(1) == 1 and dy.stride(0) % 4 == 0 and dy.data_ptr() % 16 == 0 and N % 8 == 0:
        seg = (w, 0, w.shape[1])
        if hip.linear_nn_ok(M, N, [seg]):
            return hip.linear_nn(hip.pack_quad(dy), M, N, [seg])[0]
    return hip.tile_mm(dy, hip.weight_operand(w, kmajor=True))       # (the weight's pack is shared by the step's products)


# ------------------------------------------------------------------------------- linear
TILE_LINEAR_ROWS = 64          # more rows than the skinny kernels take in one launch -> tile GEMM (0 rows = never: A/B switch)


class _Linear(torch.autograd.Function):
    """y = cat(xs) W^T + b with the concat virtual (nn.Linear over torch.cat)."""

    @staticmethod
    def forward(ctx, weight: Tensor, bias: Optional[Tensor], *xs: Tensor):
        xs = [_rows(x) for x in xs]
        M = xs[0].shape[0]
        segs, k0 = [], 0
        for x in xs:
            segs.append({"x": x, "w": weight[:, k0:k0 + x.shape[1]]})
            k0 += x.shape[1]
        assert k0 == weight.shape[1], (k0, weight.shape)
        if M > TILE_LINEAR_ROWS and weight.is_contiguous():
            # many rows (a layer applied to all T steps at once): ONE pass over the weights on the tile GEMM; the skinny
            # kernel would re-stream them per 64-row slab
            x = torch.cat(xs, 1) if len(xs) > 1 else xs[0]
            y = hip.tile_mm(x, hip.weight_operand(weight), bias=bias)
        else:
            y = hip.linear_fwd(segs, bias, M, weight.shape[0])
        ctx.save_for_backward(weight, bias, *xs)
        ctx.has_bias = bias is not None
        ctx.key = ("linear", weight.data_ptr())
        if weight.requires_grad:
            _BATCHER.note_use(ctx.key)
        return y

    @staticmethod
    def backward(ctx, dy):
        weight, bias, *xs = ctx.saved_tensors
        return _linear_backward(ctx, weight, bias, xs, dy.contiguous(), ctx.needs_input_grad)


def _linear_backward(ctx, weight, bias, xs, dy, ni):
    """autograd of y = cat(xs) W^T + b given dy: (d_w, d_b, *d_xs); dW deferred over all uses of the weight (_BATCHER).
    ni: needs-grad flags in the order (weight, bias, *xs); ctx carries key / has_bias."""
    d_w = d_b = None
    if ni[0]:
        x = torch.cat(xs, 1) if len(xs) > 1 else xs[0]

        def flush(items):
            D = torch.cat([i[0] for i in items], 0) if len(items) > 1 else items[0][0]
            X = torch.cat([i[1] for i in items], 0) if len(items) > 1 else items[0][1]
            # (into the gradient owner's buffer when it hands one out: no AccumulateGrad add, see GRAD_SINKS)
            dw = _into_grad(weight, lambda out: hip.tile_mm(D, X, a_kmajor=True, b_kmajor=True, out=out))
            db = _into_grad(bias, lambda out: hip.col_sum(D, out)) if ctx.has_bias else None
            return dw, db

        res = _BATCHER.add(ctx.key, (dy, x), (weight, bias), flush)
        if res is not None:
            d_w, d_b = res
            if not (ctx.has_bias and ni[1]):
                d_b = None
    elif ctx.has_bias and ni[1]:
        d_b = hip.col_sum(dy)
    d_xs, k0 = [], 0
    for i, x in enumerate(xs):
        k = x.shape[1]
        d_xs.append(_mm_nn(dy, weight[:, k0:k0 + k]) if ni[2 + i] else None)
        k0 += k
    return (d_w, d_b, *d_xs)


class _VocabHeadNLL(torch.autograd.Function):
    """F.log_softmax(logit(x)) + masked NLL sum + per-row argmax (captioner.py:266, :313, :361; misc/utils.py:132-146, 181-192) with
    the criterion folded into the finishing pass of the head's GEMM: the [M, V] logits are never written; what is kept for the
    backward is pre = w * (softmax - onehot), which the backward scales by the upstream scalar and feeds to nn.Linear's products."""

    @staticmethod
    def compute(x, weight, bias, target, w):
        """the forward's arithmetic: -> (loss [1], argmax [M], pre [M, V])"""
        parts = hip.tile_mm(_rows(x), hip.weight_operand(weight), parts_only=True)
        return hip.vocab_head_nll_fwd(parts, bias, target.contiguous(), w.contiguous())

    @staticmethod
    def forward(ctx, x, weight, bias, target, w, done):
        # done: compute()'s result on the same x, formed earlier (the argmax was needed before x's final graph node existed:
        # cvc.train_loops._Loop, joint back-propagation) -- this node then only attaches the backward
        x = _rows(x)
        loss, amax, pre = done if done is not None else _VocabHeadNLL.compute(x, weight, bias, target, w)
        ctx.save_for_backward(weight, bias, x, pre)
        ctx.has_bias = bias is not None
        ctx.key = ("linear", weight.data_ptr())
        if weight.requires_grad:
            _BATCHER.note_use(ctx.key)
        ctx.mark_non_differentiable(amax)
        return loss, amax

    @staticmethod
    def backward(ctx, g, _g_amax):
        weight, bias, x, pre = ctx.saved_tensors
        dy = hip.scale_by_scalar(pre, g.contiguous().reshape(1))
        ni = ctx.needs_input_grad
        d_w, d_b, d_x = _linear_backward(ctx, weight, bias, [x], dy, (ni[1], ni[2], ni[0]))
        return d_x, d_w, d_b, None, None, None


def vocab_head_nll(x: Tensor, weight: Tensor, bias: Optional[Tensor], target: Tensor, w: Tensor, done=None) -> Tuple[Tensor, Tensor]:
    """-> (sum_m w[m] * -log_softmax(x W^T + b)[m, target[m]] as a [1] tensor, argmax over V per row [M] int64); x [M, K]
    done: vocab_head_nll_compute()'s result for the same x"""
    return _VocabHeadNLL.apply(x, weight, bias, target.reshape(-1), w.reshape(-1), done)


def vocab_head_nll_compute(x: Tensor, weight: Tensor, bias: Optional[Tensor], target: Tensor, w: Tensor):
    """the arithmetic of vocab_head_nll without its graph node (-> an opaque `done` for vocab_head_nll; done[1] is the argmax)"""
    with torch.no_grad():
        return _VocabHeadNLL.compute(x.detach(), weight.detach(), None if bias is None else bias.detach(), target.reshape(-1), w.reshape(-1))


def vocab_head_nll_ok(x: Tensor, weight: Tensor) -> bool:
    return bool(x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and weight.is_contiguous() and weight.shape[0] <= 8192
                and weight.shape[0] % 4 == 0 and x.shape[0] > TILE_LINEAR_ROWS)


def linear(xs, weight: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    if isinstance(xs, Tensor):
        xs = [xs]
    lead = xs[0].shape[:-1]
    if len(lead) != 1:
        xs = [x.reshape(-1, x.shape[-1]) for x in xs]
    y = _Linear.apply(weight, bias, *xs)
    return y if len(lead) == 1 else y.reshape(*lead, -1)


# ------------------------------------------------------------------------------- LSTM cell
PACKED_LSTM_FORWARD = True     # False: the in-place (row-major weights) ring kernel for every forward (A/B switch)


def new_step() -> None:
    """Start of a training forward: weight packs derived from the parameters are rebuilt at their next use."""
    hip.lstm_train_new_step()


HOIST = True      # training loops: multiply the input segments known for all T steps once, before the loop (A/B switch)


class _HoistedGates(torch.autograd.Function):
    """G[t] = sum_s x_s[t] W_ih[:, cols_s]^T for every step t at once -- the input segments of a training loop's LSTM cell that do
    not depend on the recurrence (the embedded teacher-forced words, fc_feats, the localized context of the reconstruction
    loop; model/captioner.py:243-264, 348-360 feed them step by step).  One dense product over T * B rows instead of T passes over
    those weight columns.  The weight matrix is NOT a differentiable input here: its gradient, columns of these segments
    included, is the cell's T-batched dW product (the cell keeps stashing every segment), so autograd sees one producer."""

    @staticmethod
    def forward(ctx, w_ih, cols, per_step, T, B, *xs):
        # xs[s]: [T * B, k] (rows t * B + b) when per_step[s], else [B, k] (the same every step)
        prod = lambda x, c0, n: (hip.tile_mm(_c(x), w_ih[:, c0:c0 + n]) if x.shape[0] > 64 else
                                 hip.linear_fwd([{"x": _c(x), "w": w_ih[:, c0:c0 + n]}], None, x.shape[0], w_ih.shape[0]))
        G = None
        for x, (c0, n), ps in zip(xs, cols, per_step):
            if ps:
                g = prod(x, c0, n)
                G = g if G is None else G.add_(g)
        if G is None:
            G = w_ih.new_zeros(T * B, w_ih.shape[0])
        G = G.view(T, B, -1)
        for x, (c0, n), ps in zip(xs, cols, per_step):
            if not ps:
                G.add_(prod(x, c0, n).unsqueeze(0))
        ctx.save_for_backward(w_ih)
        ctx.cols, ctx.per_step, ctx.T, ctx.B = cols, per_step, T, B
        ctx.set_materialize_grads(False)
        return G

    @staticmethod
    def backward(ctx, dG):
        (w_ih,) = ctx.saved_tensors
        n_in = 5 + len(ctx.cols)
        if dG is None:
            return (None,) * n_in
        T, B = ctx.T, ctx.B
        dG = _c(dG).view(T * B, -1)
        d_xs = []
        for i, ((c0, n), ps) in enumerate(zip(ctx.cols, ctx.per_step)):
            if not ctx.needs_input_grad[5 + i]:
                d_xs.append(None)
            elif ps:
                d_xs.append(hip.tile_mm(dG, w_ih[:, c0:c0 + n], b_kmajor=True) if T * B > 64 else _mm_nn(dG, w_ih[:, c0:c0 + n]))
            else:                                                   # a segment shared by all steps: the steps' gradients add up
                d_xs.append(_mm_nn(dG.view(T, B, -1).sum(0), w_ih[:, c0:c0 + n]))
        return (None, None, None, None, None, *d_xs)


def hoisted_gates(w_ih: Tensor, segs, T: int, B: int):
    """segs: ((x, col0), ...) with x [B, k] (constant over the steps) or [B, T, k]; -> a list of T tensors [B, 4R], or None when
    hoisting is off / not applicable (the cell then multiplies every segment itself)."""
    if not (HOIST and w_ih.is_cuda and w_ih.is_contiguous() and PACKED_LSTM_FORWARD and T >= 1 and w_ih.shape[0] % 32 == 0):
        return None
    xs, cols, per_step = [], [], []
    for x, c0 in segs:
        ps = x.dim() == 3
        if ps:
            x = x.transpose(0, 1).reshape(T * B, x.shape[2])        # rows t * B + b: G[t] is then a contiguous [B, 4R] slice
        if x.shape[1] % 4 != 0 or c0 % 4 != 0 or x.dtype != torch.float32:
            return None
        xs.append(x)
        cols.append((int(c0), int(x.shape[1])))
        per_step.append(ps)
    G = _HoistedGates.apply(w_ih, tuple(cols), tuple(per_step), int(T), int(B), *xs)
    return list(G.unbind(0))


class _LstmCell(torch.autograd.Function):
    """nn.LSTMCell over a virtual concat of input segments (decoder_core.py:45-50, 59-61).  gate_pre / hoisted: the segments flagged
    in `hoisted` were multiplied for all steps beforehand (_HoistedGates) and arrive summed in gate_pre [M, 4R]; the launch then
    streams only the other columns of weight_ih (the pack `wp` covers exactly those) and weight_hh."""

    @staticmethod
    def forward(ctx, w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, wp, copies, gate_pre, hoisted, drop, *xs):
        xs = [_rows(x) for x in xs]
        h_prev, c_prev = _c(h_prev), _c(c_prev)
        k0 = sum(x.shape[1] for x in xs)
        assert k0 == w_ih.shape[1], ("LSTM input width mismatch", k0, tuple(w_ih.shape))
        need_bwd = any(ctx.needs_input_grad)
        M, R = c_prev.shape
        ctx.hoisted = tuple(hoisted) if gate_pre is not None else tuple(False for _ in xs)
        rec = [x for x, hz in zip(xs, ctx.hoisted) if not hz]
        # drop = (generator state, site id, p): one more output, nn.Dropout(h') with the mask generated inside the cell's kernel
        hd = None
        if wp is not None and (gate_pre is not None or all(x.data_ptr() % 16 == 0 for x in (*xs, h_prev))):
            # the decode engine's packed gate GEMM; `wp` is the pack lstm_cell() found on (or built for) the weight tensors
            # (a hoisted cell streams only the segments that were not multiplied beforehand)
            h, c, gates = hip.lstm_cell_train_fwd(rec if gate_pre is not None else xs, h_prev, c_prev, wp, b_ih, b_hh,
                                                  want_gates=need_bwd, copies=copies, gate_pre=_c(gate_pre), drop=drop)
            if drop is not None:
                h, hd = h
            hs = h if copies > 1 else (h,)
        else:
            assert gate_pre is None, "a hoisted cell runs on the packed gate GEMM"
            segs, k0 = [], 0
            for x in xs:
                segs.append({"x": x, "w": w_ih[:, k0:k0 + x.shape[1]]})
                k0 += x.shape[1]
            segs.append({"x": h_prev, "w": w_hh})
            h, c, gates = hip.lstm_cell_fwd(segs, b_ih, b_hh, c_prev, want_gates=need_bwd)
            hs = (h,) + tuple(h.clone() for _ in range(copies - 1))
            if drop is not None:
                hd = hip.dropout_rng(h, *drop)
        ctx.ncopy = len(hs)
        ctx.drop = drop
        ctx.set_materialize_grads(False)          # an unused h or c arrives as None, not as a zero-filled tensor
        if need_bwd:
            ctx.save_for_backward(w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, c, gates, *xs)
            ctx.key = ("lstm", w_ih.data_ptr())
            ctx.defer = bool(ctx.needs_input_grad[0] and ctx.needs_input_grad[1])
            if ctx.defer:
                _BATCHER.note_use(ctx.key)
        return (*hs, c) if drop is None else (*hs, hd, c)

    @staticmethod
    def backward(ctx, *d):
        # d = (gradients of the `ncopy` copies of h'..., [of the dropped copy,] d_c): summed inside the gate-gradient kernel
        w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, c_new, gates, *xs = ctx.saved_tensors
        ni = ctx.needs_input_grad
        d_hs = [g for g in d[:ctx.ncopy] if g is not None]
        d_hd = None
        if ctx.drop is not None:
            d_hd, d_c = d[ctx.ncopy], d[ctx.ncopy + 1]
        else:
            d_c = d[ctx.ncopy]
        if not d_hs and d_c is None and d_hd is None:
            d_hs = [torch.zeros_like(c_new)]      # (keeps the deferred-weight-gradient use count exact)
        if ctx.drop is not None:
            # the dropped copy's gradient rides in the third slot, where the kernel multiplies it by the regenerated mask
            assert len(d_hs) <= 2
            d_hs = [_c(g) for g in d_hs] + [None] * (2 - len(d_hs)) + [_c(d_hd) if d_hd is not None else None]
        else:
            d_hs = [_c(g) for g in d_hs] + [None] * (3 - len(d_hs))
        M, K = gates.shape
        # dX ranges that need a gradient: (weight, first column, width) in the order h_prev, xs... (a hoisted segment's gradient
        # comes from _HoistedGates' one product over all steps, not from here)
        XS0 = 11                                                      # index of xs[0] among forward's inputs
        want_x = [bool(ni[XS0 + i]) and not ctx.hoisted[i] for i in range(len(xs))]
        ranges, k0 = ([(w_hh, 0, w_hh.shape[1])] if ni[4] else []), 0
        for i, x in enumerate(xs):
            if want_x[i]:
                ranges.append((w_ih, k0, x.shape[1]))
            k0 += x.shape[1]
        use_nn = bool(ranges) and hip.linear_nn_ok(M, K, ranges)
        pw = hip.lstm_pointwise_bwd(d_hs[0], _c(d_c), gates, c_prev, c_new, want_quad=use_nn, d_h2=d_hs[1], d_h3=d_hs[2],
                                    drop3=ctx.drop)
        d_gates, d_c_prev = pw[0], pw[1]
        d_w_ih = d_w_hh = d_b = None
        if ctx.defer:
            def flush(items):
                # one cat per operand over all stashed steps (the input segments are only joined here, once)
                D = torch.cat([i[0] for i in items], 0) if len(items) > 1 else items[0][0]
                Hp = torch.cat([i[1] for i in items], 0) if len(items) > 1 else items[0][1]
                nseg = len(items[0]) - 2
                cols = [torch.cat([i[2 + k] for i in items], 0) if len(items) > 1 else items[0][2 + k] for k in range(nseg)]
                X = torch.cat(cols, 1) if nseg > 1 else cols[0]
                db = D.sum(0)
                Dp = hip.TileOperand(D, kmajor=True)                    # dY^T packed once for both weight matrices
                return hip.tile_mm(Dp, X, b_kmajor=True), hip.tile_mm(Dp, Hp, b_kmajor=True), db, db

            res = _BATCHER.add(ctx.key, (d_gates, h_prev, *xs), (w_ih, w_hh, b_ih, b_hh), flush)
            if res is not None:
                d_w_ih, d_w_hh, d_b, _ = res
        else:
            d_w_ih = _mm_tn(d_gates, torch.cat(xs, 1) if len(xs) > 1 else xs[0]) if ni[0] else None
            d_w_hh = _mm_tn(d_gates, h_prev) if ni[1] else None
            d_b = d_gates.sum(0) if (ni[2] or ni[3]) else None
        if use_nn:                      # every needed dX from one pass over the weights (csrc/gemm_nn.hip)
            got = iter(hip.linear_nn(pw[2], M, K, ranges))
            d_h_prev = next(got) if ni[4] else None
            d_xs = [next(got) if want_x[i] else None for i in range(len(xs))]
        else:                           # widths the kernel does not take (not multiples of 4): library GEMM
            d_h_prev = _mm_nn(d_gates, w_hh) if ni[4] else None
            d_xs, k0 = [], 0
            for i, x in enumerate(xs):
                k = x.shape[1]
                d_xs.append(_mm_nn(d_gates, w_ih[:, k0:k0 + k]) if want_x[i] else None)
                k0 += k
        return (d_w_ih, d_w_hh, d_b if ni[2] else None, d_b if ni[3] else None, d_h_prev,
                d_c_prev if ni[5] else None, None, None, (d_gates if ni[8] else None), None, None, *d_xs)


def lstm_cell(xs: Sequence[Tensor], h_prev: Tensor, c_prev: Tensor, w_ih, w_hh, b_ih, b_hh, copies: int = 1,
              gate_pre: Optional[Tensor] = None, hoisted: Optional[Sequence[bool]] = None, drop=None):
    """-> (h', c'), or with copies = k > 1: (h'_1, ..., h'_k, c') -- k tensors holding the same h', one per consumer, so that
    autograd has no fan-out to accumulate (their gradients are summed inside the cell's backward kernel).
    gate_pre [M, 4R] + hoisted (one flag per segment of xs): this step's slice of hoisted_gates() for the flagged segments.
    drop = (generator state, site id, p) (cvc/dropout.py): -> (h'_1, ..[h'_2], dropout(h'), c'), copies <= 2 -- the cell's kernel
    writes nn.Dropout(h') as one more tensor, mask generated in the kernel (decoder_core.py:62, 109)."""
    wp = None
    M, R = c_prev.shape
    if drop is not None:
        copies = min(2, int(copies))
    packable = PACKED_LSTM_FORWARD and w_ih.is_cuda and w_ih.is_contiguous() and w_hh.is_contiguous()
    if gate_pre is not None:
        widths = [x.shape[1] for x, hz in zip(xs, hoisted) if not hz] + [h_prev.shape[1]]
        ok = packable and hip.lstm_train_ok(M, R, widths) and all(x.data_ptr() % 16 == 0 for x in (*xs, h_prev))
        if not ok:
            # shapes the packed kernel does not take: the plain cell over ALL segments (xs holds every one of them); gate_pre is
            # ignored, so its producer receives no gradient and the hoisted segments get theirs from this cell instead
            return lstm_cell(xs, h_prev, c_prev, w_ih, w_hh, b_ih, b_hh, copies=copies, drop=drop)
        cols, k0 = [], 0
        for x, hz in zip(xs, hoisted):
            if not hz:
                cols.append((k0, x.shape[1]))
            k0 += x.shape[1]
        wp = hip.lstm_train_pack(w_ih, w_hh, cols=cols)
        return _LstmCell.apply(w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, wp, max(1, min(3, int(copies))), gate_pre, tuple(hoisted), drop,
                               *xs)
    if packable and hip.lstm_train_ok(M, R, [x.shape[1] for x in xs] + [h_prev.shape[1]]):
        wp = hip.lstm_train_pack(w_ih, w_hh)          # lives on the parameter object; rebuilt once per optimizer step
    return _LstmCell.apply(w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, wp, max(1, min(3, int(copies))), None, None, drop, *xs)


# ------------------------------------------------------------------------------- attention
class _Attention(torch.autograd.Function):
    """Soft attention over 1 or 2 feature sets sharing the query (modules.py:24-159).

    outputs: ctx_total [rows,R] (sum over sets), then per set: ctx_s, attn_s, frame_masked_s
    (frame_masked_s is a dummy 0-d tensor when no proposal_frame_mask was given)."""

    @staticmethod
    def forward(ctx, kind: int, inv_temp: float, nq: int, masks, q, w_a, b_a, *feats):
        nsets = len(feats) // 2
        q = _c(q)
        nclip = feats[0].shape[0]
        sets = []
        for s in range(nsets):
            m, fmk = masks[s][:2]
            sets.append({"proj": _c(feats[2 * s]), "ctx": _c(feats[2 * s + 1]), "mask": m, "frame_mask": fmk,
                         "sentinel": len(masks[s]) > 2 and bool(masks[s][2])})
        w_flat = None if w_a is None else w_a.reshape(-1)
        outs, ctx_sum = hip.attn_fwd(kind, q, w_flat, b_a, inv_temp, sets, nclip, nq, want_ctx=True, want_sum=nsets > 1)
        ctx.kind, ctx.inv_temp, ctx.nq, ctx.nsets, ctx.nclip = kind, inv_temp, nq, nsets, nclip
        ctx.save_for_backward(q, w_a, *[t for s in sets for t in (s["proj"], s["ctx"])], *[o[2] for o in outs])
        ctx.set_materialize_grads(False)
        # alpha_net's weight / bias gradients are row sums over every step's partials: defer them like the GEMM weights
        ctx.defer = bool(kind == hip.ATTN_ADDITIVE and w_a is not None and w_a.requires_grad and b_a is not None
                         and b_a.requires_grad)
        ctx.b_a_ref = b_a
        if ctx.defer:
            ctx.key = ("attn", w_a.data_ptr())
            _BATCHER.note_use(ctx.key)
        res = [ctx_sum if nsets > 1 else outs[0][3]]
        non_diff = []
        for (scores, fm, attn, ctx_out) in outs:
            fm_out = fm if fm is not None else q.new_empty(())      # placeholder, never read (no fill launch)
            res += [ctx_out, attn, fm_out]
            non_diff.append(attn)
        ctx.mark_non_differentiable(*non_diff)
        return tuple(res)

    @staticmethod
    def backward(ctx, d_total, *d_outs):
        q, w_a, *rest = ctx.saved_tensors
        nsets = ctx.nsets
        feats, attns = rest[:2 * nsets], rest[2 * nsets:]
        ni = ctx.needs_input_grad   # (kind, inv_temp, nq, masks, q, w_a, b_a, *feats)
        w_flat = None if w_a is None else w_a.reshape(-1)
        d_q = None
        d_w = None
        d_b = None
        d_feats: List[Optional[Tensor]] = []
        stash_w, stash_b = [], []
        for s in range(nsets):
            d_ctx_s, _d_attn, d_fm = d_outs[3 * s], d_outs[3 * s + 1], d_outs[3 * s + 2]
            d_ctx = d_ctx_s
            if d_total is not None:
                d_ctx = d_total if d_ctx is None else d_ctx + d_total
            if d_fm is not None and d_fm.dim() == 0:
                d_fm = None
            if d_ctx is None and d_fm is None:
                d_feats += [None, None]
                continue
            proj, cfeat = feats[2 * s], feats[2 * s + 1]
            want_dp, want_dc = ni[7 + 2 * s], ni[8 + 2 * s]
            same = proj.data_ptr() == cfeat.data_ptr()
            d_scores, d_q_s, d_w_part, d_proj, d_cf = hip.attn_bwd(
                ctx.kind, q, w_flat, ctx.inv_temp, proj, cfeat, attns[s], _c(d_ctx), _c(d_fm), ctx.nclip, ctx.nq,
                want_dp, want_dc, ni[5])
            d_q = d_q_s if d_q is None else d_q + d_q_s
            if ctx.kind == hip.ATTN_ADDITIVE and ctx.defer:
                stash_w.append(d_w_part)
                stash_b.append(d_scores.reshape(-1))
            elif ctx.kind == hip.ATTN_ADDITIVE:
                if ni[5]:
                    dw = d_w_part.sum(0)
                    d_w = dw if d_w is None else d_w + dw
                if ni[6]:
                    db = d_scores.sum().reshape(1)
                    d_b = db if d_b is None else d_b + db
            if same and d_proj is not None and d_cf is not None:
                d_proj = d_proj + d_cf
                d_cf = None
            d_feats += [d_proj, d_cf]
        if ctx.kind == hip.ATTN_ADDITIVE and ctx.defer:
            def flush(items):
                dw = torch.cat([t for it in items for t in it[0]], 0).sum(0).reshape(w_a.shape)
                db = torch.cat([t for it in items for t in it[1]], 0).sum().reshape(1)
                return dw, db

            got = _BATCHER.add(ctx.key, (stash_w, stash_b), (w_a, ctx.b_a_ref), flush)
            if got is not None:
                d_w, d_b = got
        elif d_w is not None:
            d_w = d_w.reshape(w_a.shape)
        return (None, None, None, None, d_q if ni[4] else None, d_w, d_b, *d_feats)


def attention(kind: int, q: Tensor, w_a: Optional[Tensor], b_a: Optional[Tensor], inv_temp: float,
              sets: Sequence[Tuple[Tensor, Tensor, Optional[Tensor], Optional[Tensor]]], with_sentinel: bool = False):
    """sets: (proj_context, context, mask, proposal_frame_mask) per feature set.  with_sentinel: masked positions are filled
    with -inf instead of -1e8 (modules.py:40-41, 123-124).
    Returns (ctx_total, [(ctx_s, attn_s, frame_masked_s or None)])."""
    nclip = sets[0][0].shape[0]
    nq = q.shape[0] // nclip
    assert nq * nclip == q.shape[0], "query rows must be a multiple of the number of clips"
    ws = list(with_sentinel) if isinstance(with_sentinel, (list, tuple)) else [with_sentinel] * len(sets)      # per set, or one for all
    masks = tuple((s[2], s[3], bool(w)) for s, w in zip(sets, ws))
    feats = [t for s in sets for t in (s[0], s[1])]
    res = _Attention.apply(kind, float(inv_temp), nq, masks, q, w_a, b_a, *feats)
    out = []
    for i, s in enumerate(sets):
        c, a, fm = res[1 + 3 * i], res[2 + 3 * i], res[3 + 3 * i]
        out.append((c, a, fm if s[3] is not None else None))
    return res[0], out


# ------------------------------------------------------------------------------- embedding / vocab head
class _EmbedRelu(torch.autograd.Function):
    @staticmethod
    def forward(ctx, table, idx, drop, rng):
        idx = idx.contiguous()
        # rng = (generator state, site id, p): the keep-mask is generated inside the kernel (cvc/dropout.py), forward and backward
        out = hip.embed_relu_rng_fwd(table, idx, *rng) if rng is not None else hip.embed_relu_fwd(table, idx, drop)
        ctx.save_for_backward(table, idx, drop if drop is not None else table.new_empty(()))      # (placeholder, never read)
        ctx.has_drop = drop is not None
        ctx.rng = rng
        ctx.key = ("embed", table.data_ptr())
        if table.requires_grad:
            _BATCHER.note_use(ctx.key)
        return out

    @staticmethod
    def backward(ctx, d_out):
        table, idx, drop = ctx.saved_tensors
        item = (idx, drop if ctx.has_drop else None, ctx.rng, d_out.contiguous())

        def flush(items):
            # every lookup of the table in this pass (the embedded words of loops A, B, C) accumulates into ONE buffer -- the
            # gradient owner's when it hands it out (GRAD_SINKS), else a fresh zero-filled one: one fill, no framework adds
            buf, sink = claim_grad(table)
            d_table = buf if buf is not None else torch.zeros_like(table)
            for i_idx, i_drop, i_rng, i_d in items:
                if i_rng is not None:
                    hip.embed_relu_rng_bwd(table, i_idx, *i_rng, i_d, d_table=d_table)
                else:
                    hip.embed_relu_bwd(table, i_idx, i_drop, i_d, d_table=d_table)
            if sink is not None:
                sink.written(table)
                return (None,)
            return (d_table,)

        got = _BATCHER.add(ctx.key, item, (table,), flush)
        return (got[0] if got is not None else None), None, None, None


def embed_relu(table: Tensor, idx: Tensor, drop: Optional[Tensor] = None, rng=None) -> Tensor:
    """relu(Embedding(idx)) (* dropout keep-mask / (1-p)), captioner.py:53-68.  drop: the mask as a tensor; rng = (state, site,
    p): the mask generated in the kernel instead (one or the other)."""
    assert drop is None or rng is None
    shape = idx.shape
    out = _EmbedRelu.apply(table, idx.reshape(-1), drop, rng)
    return out.reshape(*shape, -1)


class _LogSoftmax(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = hip.log_softmax_fwd(x.contiguous())
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return hip.log_softmax_bwd(y, dy.contiguous())


def log_softmax(x: Tensor) -> Tensor:
    """F.log_softmax(x, dim=-1) for [M, V]."""
    lead = x.shape[:-1]
    y = _LogSoftmax.apply(x.reshape(-1, x.shape[-1]))
    return y.reshape(*lead, -1)


class _MaskedNLLSum(torch.autograd.Function):
    """sum_m w[m] * -logp[m, target[m]]  (misc/utils.py:139-146 before the mean)."""

    @staticmethod
    def forward(ctx, logp, target, w):
        logp, target, w = logp.contiguous(), target.contiguous(), w.contiguous()
        ctx.save_for_backward(target, w)
        ctx.V = logp.shape[1]
        return hip.nll_fwd(logp, target, w)

    @staticmethod
    def backward(ctx, g):
        target, w = ctx.saved_tensors
        return hip.nll_bwd(target, w, g.contiguous().reshape(1), ctx.V), None, None


def masked_nll_sum(logp: Tensor, target: Tensor, w: Tensor) -> Tensor:
    return _MaskedNLLSum.apply(logp, target.reshape(-1), w.reshape(-1))


class _VocabNLL(torch.autograd.Function):
    """Masked NLL sum + per-row argmax straight from raw logits: log_softmax, the NLL gather and the cycle's
    argmax cut (captioner.py:266, :313; misc/utils.py:139-146) in one pass, no [M, V] log-prob matrix."""

    @staticmethod
    def forward(ctx, logits, target, w):
        logits, target, w = logits.contiguous(), target.contiguous(), w.contiguous()
        loss, lse, amax = hip.vocab_nll_fwd(logits, target, w)
        ctx.save_for_backward(logits, lse, target, w)
        ctx.mark_non_differentiable(amax)
        return loss, amax

    @staticmethod
    def backward(ctx, g, _g_amax):
        logits, lse, target, w = ctx.saved_tensors
        return hip.vocab_nll_bwd(logits, lse, target, w, g.contiguous().reshape(1)), None, None


def vocab_nll(logits: Tensor, target: Tensor, w: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (sum_m w[m] * -log_softmax(logits)[m, target[m]]  as a [1] tensor,  argmax over V per row [M] int64)."""
    return _VocabNLL.apply(logits, target.reshape(-1), w.reshape(-1))


class _Grounder(torch.autograd.Function):
    @staticmethod
    def forward(ctx, xt, feats, bias, mask):
        xt, feats = xt.contiguous(), feats.contiguous()
        bias = None if bias is None else bias.contiguous()
        out = hip.grounder_fwd(xt, feats, bias, mask)
        ctx.save_for_backward(xt, feats, mask if mask is not None else xt.new_zeros((), dtype=torch.bool))
        ctx.has_mask, ctx.has_bias = mask is not None, bias is not None
        return out

    @staticmethod
    def backward(ctx, d_out):
        xt, feats, mask = ctx.saved_tensors
        # captioner.py:171 masks with an autograd-visible masked_fill_: no gradient at filled slots
        d = (d_out.masked_fill(mask, 0) if ctx.has_mask else d_out).contiguous()
        ni = ctx.needs_input_grad
        d_xt, d_feats = hip.grounder_bwd(d, xt, feats, ni[0], ni[1]) if (ni[0] or ni[1]) else (None, None)
        return d_xt, d_feats, d if (ctx.has_bias and ni[2]) else None, None


def grounder(xt: Tensor, feats: Tensor, bias: Optional[Tensor], mask: Optional[Tensor]) -> Tensor:
    """captioner.py:132-173, dot-product branch."""
    return _Grounder.apply(xt, feats, bias, mask)


class _AttnNLL(torch.autograd.Function):
    """att2_loss and ground_loss (misc/utils.py:150-162) of all T steps in one launch pair: -mean over the labelled proposals of
    log_softmax(scores, 2), 0 when nothing is labelled (the reference branches on the host; here max(count, 1))."""

    @staticmethod
    def forward(ctx, x0, x1, target):
        loss, ws, tg = hip.attn_nll_fwd(x0, x1, target)
        ctx.save_for_backward(x0, x1, tg, ws)
        ctx.set_materialize_grads(False)        # an unused loss (ground_loss is never optimised, trainer.py:93-95) arrives as None
        return loss[0:1], loss[1:2]

    @staticmethod
    def backward(ctx, g0, g1):
        x0, x1, tg, ws = ctx.saved_tensors
        ni = ctx.needs_input_grad
        g0 = None if g0 is None else g0.contiguous()
        g1 = None if g1 is None else g1.contiguous()
        want0, want1 = ni[0] and g0 is not None, ni[1] and g1 is not None
        if not (want0 or want1):
            return None, None, None
        d0, d1 = hip.attn_nll_bwd(x0, x1, tg, ws, g0, g1, want0, want1)
        return d0, d1, None


def attn_nll(att2_weights: Tensor, ground_weights: Tensor, target: Tensor) -> Tuple[Tensor, Tensor]:
    """-> (att2_loss [1], ground_loss [1])"""
    return _AttnNLL.apply(att2_weights, ground_weights, target)
