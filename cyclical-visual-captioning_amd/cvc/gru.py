"""Frame-context GRU of the once-per-clip encoder on the HIP kernels (reference backbone.py:103-106 builds
`nn.GRU(R, R/2, 2, dropout=0.2, bidirectional=True, batch_first=True)`, :335-338 runs it over the F sampled frames).

The reference hands the whole recurrence to cuDNN; on ROCm the same module runs in MIOpen (79 of the encoder's 84 ms at
config 2).  Here a layer is
  1. ONE dense GEMM for the input projections of all F steps and both directions (the tile GEMM, csrc/gemm_tile.hip), and
  2. the recurrence, in one of two forms with identical results:
     - persistent (cvc_gru_seq_persistent_fwd, csrc/gru_persistent.hip; H % 128 == 0, H <= 1024): one cooperative launch
       for the whole sequence, W_hh held in registers, steps separated by a barrier in device memory;
     - per step (cvc_gru_seq_fwd, csrc/gemm_packed.hip): F launches of the packed gate-GEMM kernel in its GRU form, W_hh of
       both directions (25 MB at H = 1024) re-read from the Infinity Cache every step -- any H % 8 == 0, and the fallback
       when the persistent form cannot run or reports a barrier time-out.
`gru_forward` is the inference path (no autograd).  `gru_forward_train` is the same recurrence under autograd: the forward
additionally keeps every step's gates -- persistent form for H % 128 == 0, H <= 1024, else the per-step training form
(cvc_gru_seq_train_fwd, any H % 8 == 0: config 5's width, rnn_size 4096 -> H = 2048, is beyond what 256 CUs keep in registers:
W_hh of both directions is 151 MB as split bf16 terms against 128 MB of vector registers on the chip) -- the backward walks the
sequence backwards (persistent for H % 256 == 0, H <= 1024, else cvc_gru_seq_bwd: gate arithmetic + dgh W_hh per step on the LSTM
cells' backward-data kernel) and takes dW_ih, dW_hh, dX and the biases from dense products over all steps on the tile GEMM."""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import hip
from .decode import pack_weights

PERSISTENT = True       # False: always the per-step form (A/B switch)
last_form = None        # "persistent" / "steps": which form produced the last layer (tests, bench)


def pack_gru_weights(w_hh: torch.Tensor, H: int) -> torch.Tensor:
    """[3H, H] (r, z, n) -> the packed-LSTM block order with a zero fourth gate, columns zero-padded to a multiple of 32:
    [H/8][Kp/4][32][4]."""
    assert w_hh.shape == (3 * H, H) and H % 8 == 0
    Kp = (H + 31) // 32 * 32
    w = w_hh.new_zeros(4 * H, Kp)
    w[:3 * H, :H] = w_hh
    return pack_weights(w, lstm_R=H)


def pack_gru_weights_t(w_hh: torch.Tensor, H: int) -> torch.Tensor:
    """[3H, H] -> W_hh^T for the persistent backward: [H/8 blocks][3H/8 k groups][8 units][8 k] (a workgroup's 8 weight COLUMNS,
    8 consecutive k per lane)."""
    assert w_hh.shape == (3 * H, H) and H % 8 == 0
    return w_hh.reshape(3 * H // 8, 8, H // 8, 8).permute(2, 0, 3, 1).contiguous()


BWD_PERSISTENT = True      # False: always the per-step backward (A/B switch)
last_bwd_form = None


class GruUnavailable(RuntimeError):
    """The persistent recurrence could not run this call (grid not co-resident, barrier time-out): the caller falls back to the
    library module (cvc/model/backbone.py) instead of aborting the training run."""


def supported(gru: nn.Module, x: torch.Tensor) -> bool:
    """Shapes / modules the inference path takes."""
    return (isinstance(gru, nn.GRU) and gru.batch_first and gru.bias and gru.hidden_size % 8 == 0 and x.dim() == 3
            and x.is_cuda and x.dtype == torch.float32)


def supported_train(gru: nn.Module, x: torch.Tensor) -> bool:
    """... the autograd path takes: the same (the persistent forms where they exist, the per-step forms for every other width)."""
    return supported(gru, x)


last_train_form = None     # "persistent" / "steps": which forward form the last autograd layer ran


def _layer_operands(gru: nn.GRU):
    """Per layer: (W_ih of both directions as a tile operand, packed W_hh [ndir][...], b_ih [ndir, 3H], b_hh [ndir, 3H]);
    rebuilt when any parameter changed."""
    params = list(gru.parameters())
    # (the generation covers updates that leave _version alone: fused Adam, graph replays -- hip.bump_weights_generation)
    stamp = (hip.weights_generation(),) + tuple((p.data_ptr(), p._version) for p in params)
    ent = getattr(gru, "_cvc_gru_pack", None)          # lives on the module (no table keyed by ids / addresses)
    if ent is not None and ent[0] == stamp:
        return ent[1]
    H, sfx = gru.hidden_size, ([""] if not gru.bidirectional else ["", "_reverse"])
    layers = []
    with torch.no_grad():
        for l in range(gru.num_layers):
            g = lambda n: [getattr(gru, f"{n}_l{l}{s}").detach().float() for s in sfx]
            w_ih = torch.cat(g("weight_ih"), 0).contiguous()
            layers.append((hip.TileOperand(w_ih, kmajor=False), torch.stack([pack_gru_weights(w, H) for w in g("weight_hh")]),
                           torch.stack(g("bias_ih")).contiguous(), torch.stack(g("bias_hh")).contiguous()))
    gru._cvc_gru_pack = (stamp, layers)
    return layers


def gru_forward(gru: nn.GRU, x: torch.Tensor) -> torch.Tensor:
    """x [B, F, in] -> [B, F, ndir * H], the first output of `gru(x)` (h0 = 0)."""
    assert supported(gru, x), "shape / module outside the HIP GRU's range"
    B, F, _ = x.shape
    H, ndir = gru.hidden_size, 2 if gru.bidirectional else 1
    layers = _layer_operands(gru)
    out = torch.empty(B, F, ndir * H, device=x.device, dtype=torch.float32)
    L, st = hip.lib(), hip._stream()
    global last_form
    try_persistent = PERSISTENT and H % 128 == 0 and H <= 1024 and not torch.cuda.is_current_stream_capturing()

    def run(persistent: bool):
        """All chunks and layers in one go; returns the persistent launches' error words (device tensors, not read here)."""
        words = []
        for b0 in range(0, B, 64):
            m = min(64, B - b0)
            cur = x[b0:b0 + m].transpose(0, 1).contiguous().view(F * m, -1)            # time-major rows (t, clip)
            hq = torch.empty(2 * ndir * ((H + 31) // 32 * 32) * 64, device=x.device, dtype=torch.float32)
            for l, (w_ih, wp, b_ih, b_hh) in enumerate(layers):
                gi = hip.tile_mm(cur, w_ih)                                              # [F*m, ndir*3H], no bias
                last = l == len(layers) - 1
                if last:
                    y, ld_m, ld_t = out[b0:b0 + m], F * ndir * H, ndir * H
                else:
                    y = torch.empty(F * m, ndir * H, device=x.device, dtype=torch.float32)
                    ld_m, ld_t = ndir * H, m * ndir * H
                args = (wp.data_ptr(), gi.data_ptr(), ndir * 3 * H, m * ndir * 3 * H, b_ih.data_ptr(), b_hh.data_ptr(), m, F, H, ndir,
                        hq.data_ptr(), y.data_ptr(), ld_m, ld_t)
                if persistent:
                    sync = torch.zeros(int(L.cvc_gru_persistent_sync_words()), device=x.device, dtype=torch.int32)
                    slots = torch.empty((F + 1) * ndir * H * 64, device=x.device, dtype=torch.float32)     # one state slot per step
                    pargs = args[:10] + (slots.data_ptr(),) + args[11:]
                    if L.cvc_gru_seq_persistent_fwd(*pargs, sync.data_ptr(), st) != 0:
                        return None                                                    # launch refused (shape / residency): per-step form
                    words.append(sync[4:5])
                else:
                    hip._check(L.cvc_gru_seq_fwd(*args, st), "cvc_gru_seq_fwd")
                cur = y
        return words

    if try_persistent:
        # every layer (and 64-clip chunk) is enqueued before the error words are looked at: ONE host read per call instead of one
        # per layer; a barrier time-out anywhere leaves its word set and the whole call is redone in the per-step form
        words = run(True)
        if words is not None and int(torch.cat(words).abs().sum()) == 0:
            last_form = "persistent"
            return out
    run(False)
    last_form = "steps"
    return out


# ------------------------------------------------------------------------------------------------ training (autograd)
class _GruLayer(torch.autograd.Function):
    """One GRU layer over time-major rows (t * m + clip): x [F*m, in] -> y [F*m, ndir*H]."""

    @staticmethod
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, m, F):
        # w_ih [ndir*3H, in], w_hh [ndir, 3H, H], b_ih / b_hh [ndir, 3H]
        ndir, H = w_hh.shape[0], w_hh.shape[2]
        L, st = hip.lib(), hip._stream()
        x, w_ih, w_hh = x.contiguous(), w_ih.contiguous(), w_hh.contiguous()
        gi = hip.tile_mm(x, w_ih)
        wp = torch.stack([pack_gru_weights(w_hh[d], H) for d in range(ndir)])
        y = torch.empty(F * m, ndir * H, device=x.device, dtype=torch.float32)
        gates = torch.empty(F * m, ndir * 4 * H, device=x.device, dtype=torch.float32)
        global last_train_form
        b_ih, b_hh = b_ih.contiguous(), b_hh.contiguous()
        done = False
        if PERSISTENT and H % 128 == 0 and H <= 1024:
            sync = torch.zeros(int(L.cvc_gru_persistent_sync_words()), device=x.device, dtype=torch.int32)
            slots = torch.empty((F + 1) * ndir * H * 64, device=x.device, dtype=torch.float32)
            rc = L.cvc_gru_seq_persistent_train_fwd(wp.data_ptr(), gi.data_ptr(), ndir * 3 * H, m * ndir * 3 * H, b_ih.data_ptr(),
                                                    b_hh.data_ptr(), m, F, H, ndir, slots.data_ptr(), y.data_ptr(), ndir * H,
                                                    m * ndir * H, gates.data_ptr(), ndir * 4 * H, m * ndir * 4 * H, sync.data_ptr(), st)
            # (eagerly a host read: a barrier time-out repeats the layer in the per-step form; in deferred mode -- captured training
            # steps, cvc.hip.defer_errors -- the word is OR-ed into the step's status word and the step is re-run later if it was set)
            done = rc == 0 and hip.error_word_ok(sync[4:5])
        if not done:
            # per-step training form: any H % 8 == 0 (and the fallback of the persistent form), one launch per time step
            hq = torch.empty(2 * ndir * ((H + 31) // 32 * 32) * 64, device=x.device, dtype=torch.float32)
            hip._check(L.cvc_gru_seq_train_fwd(wp.data_ptr(), gi.data_ptr(), ndir * 3 * H, m * ndir * 3 * H, b_ih.data_ptr(), b_hh.data_ptr(),
                                               m, F, H, ndir, hq.data_ptr(), y.data_ptr(), ndir * H, m * ndir * H, gates.data_ptr(),
                                               ndir * 4 * H, m * ndir * 4 * H, st), "cvc_gru_seq_train_fwd")
        last_train_form = "persistent" if done else "steps"
        ctx.save_for_backward(x, w_ih, w_hh, gates, y)
        ctx.dims = (m, F, H, ndir)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w_ih, w_hh, gates, y = ctx.saved_tensors
        m, F, H, ndir = ctx.dims
        L, st = hip.lib(), hip._stream()
        dy = dy.contiguous()
        dgi = torch.empty(F * m, ndir * 3 * H, device=x.device, dtype=torch.float32)
        dgh = torch.empty_like(dgi)
        global last_bwd_form
        done = False
        if BWD_PERSISTENT and H % 256 == 0 and H <= 1024:
            wt = torch.stack([pack_gru_weights_t(w_hh[d], H) for d in range(ndir)])
            slots = torch.empty(F * ndir * 3 * H * 64, device=x.device, dtype=torch.float32)
            sync = torch.zeros(int(L.cvc_gru_bwd_persistent_sync_words()), device=x.device, dtype=torch.int32)
            rc = L.cvc_gru_seq_bwd_persistent(dy.data_ptr(), ndir * H, m * ndir * H, gates.data_ptr(), ndir * 4 * H, m * ndir * 4 * H,
                                              y.data_ptr(), ndir * H, m * ndir * H, wt.data_ptr(), m, F, H, ndir, dgi.data_ptr(),
                                              dgh.data_ptr(), slots.data_ptr(), sync.data_ptr(), st)
            done = rc == 0 and hip.error_word_ok(sync[4:5])
        if not done:
            ks = int(L.cvc_gru_seq_bwd_ksplit(H))
            work = torch.empty(ndir * (2 * m * H + 3 * H * 64 + ks * m * ((H + 127) // 128) * 128), device=x.device, dtype=torch.float32)
            hip._check(L.cvc_gru_seq_bwd(dy.data_ptr(), ndir * H, m * ndir * H, gates.data_ptr(), ndir * 4 * H, m * ndir * 4 * H, y.data_ptr(),
                                         ndir * H, m * ndir * H, w_hh.data_ptr(), m, F, H, ndir, dgi.data_ptr(), dgh.data_ptr(),
                                         work.data_ptr(), st), "cvc_gru_seq_bwd")
        last_bwd_form = "persistent" if done else "steps"
        ni = ctx.needs_input_grad
        Gi = hip.TileOperand(dgi, kmajor=True) if (ni[1]) else None           # dG^T packed once
        d_x = hip.tile_mm(dgi, w_ih, b_kmajor=True) if ni[0] else None          # [F*m, in]
        d_w_ih = hip.tile_mm(Gi, x, b_kmajor=True) if ni[1] else None           # [ndir*3H, in]
        d_w_hh = None
        if ni[2]:
            zeros = y.new_zeros(m, H)
            parts = []
            for d in range(ndir):
                yd = y[:, d * H:(d + 1) * H]
                hp = torch.cat((zeros, yd[:-m]), 0) if d == 0 else torch.cat((yd[m:], zeros), 0)     # h_{t-1} of this direction
                parts.append(hip.tile_mm(dgh[:, d * 3 * H:(d + 1) * 3 * H], hp, a_kmajor=True, b_kmajor=True))
            d_w_hh = torch.stack(parts)
        d_b_ih = dgi.sum(0).view(ndir, 3 * H) if ni[3] else None
        d_b_hh = dgh.sum(0).view(ndir, 3 * H) if ni[4] else None
        return d_x, d_w_ih, d_w_hh, d_b_ih, d_b_hh, None, None


def gru_forward_train(gru: nn.GRU, x: torch.Tensor) -> torch.Tensor:
    """`gru(x)[0]` under autograd on the HIP kernels (h0 = 0; inter-layer dropout as the module has it)."""
    assert supported_train(gru, x), "shape / module outside the HIP GRU's autograd range"
    B, F, _ = x.shape
    H, ndir = gru.hidden_size, 2 if gru.bidirectional else 1
    sfx = [""] if not gru.bidirectional else ["", "_reverse"]
    outs = []
    for b0 in range(0, B, 64):
        m = min(64, B - b0)
        cur = x[b0:b0 + m].transpose(0, 1).reshape(F * m, -1)                       # time-major rows (t, clip)
        for l in range(gru.num_layers):
            g = lambda n: [getattr(gru, f"{n}_l{l}{s}") for s in sfx]
            cur = _GruLayer.apply(cur, torch.cat(g("weight_ih"), 0), torch.stack(g("weight_hh")), torch.stack(g("bias_ih")),
                                  torch.stack(g("bias_hh")), m, F)
            if gru.training and gru.dropout > 0 and l + 1 < gru.num_layers:
                # nn.GRU's inter-layer dropout with the mask of site enc.gru.<l> from the in-kernel generator (cvc/dropout.py)
                from . import dropout as _dropout
                cur = _dropout.apply_p(cur, gru.dropout, "enc.gru.%d" % l)
        outs.append(cur.view(F, m, ndir * H).transpose(0, 1))
    return torch.cat(outs, 0) if len(outs) > 1 else outs[0].contiguous()
