"""CPU oracle for the caption-decode hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this file;
the product package (cyclical-visual-captioning_amd/cvc) never does and fails loudly
when its HIP library is missing.

This is a functional, fp32, torch-CPU restatement of the reference algorithm (the arithmetic
is floating point, so the oracle keeps the reference's own ATen op sequence: linear, tanh,
softmax, bmm, lstm_cell, log_softmax, topk -- that also makes it the honest CPU baseline,
SURVEY.md section 8(d)).  Every function cites the reference file:line it follows
(paths relative to /root/reference/anet-video-captioning/).

Parity pin: tests/test_oracle_golden.py checks every function here against vectors captured
by importing the reference itself in the build container (tools/make_golden.py ->
tests/golden/*.npz).  The reference has no tests / golden vectors of its own.
Beam search and the sharded-gradient mean have no reference implementation at all; they are
build-defined and pinned as SURVEY.md section 8(c)(iv,v) describes ("parity unpinned by the
reference" for those two).
"""
from __future__ import annotations

import itertools
import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
MIN_VALUE = -1e8  # model/modules.py:22,98 -- finite, so an all-masked row is uniform, not NaN


# ----------------------------------------------------------------------------- attention
def additive_attention(h: Tensor, p_ctx: Tensor, ctx: Optional[Tensor], mask: Optional[Tensor],
                       frame_mask: Optional[Tensor], w_h: Tensor, b_h: Tensor, w_a: Tensor, b_a: Tensor,
                       with_sentinel: bool = False) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """model/modules.py:100-159 (AdditiveSoftAttention.forward); no temperature (:120).  with_sentinel: -inf fill (:123-124,
    :136-138) instead of -1e8."""
    MIN_VALUE = -float("inf") if with_sentinel else globals()["MIN_VALUE"]
    n = p_ctx.size(1)
    q = F.linear(h, w_h, b_h).unsqueeze(1)                       # :109-110
    e = torch.tanh(p_ctx + q)                                    # :111-112
    s = F.linear(e.view(-1, e.size(2)), w_a, b_a).view(-1, n)    # :113-115
    if mask is not None:
        s.data.masked_fill_(mask, MIN_VALUE)                     # :129 (on .data: invisible to autograd)
    fm = None
    if frame_mask is not None:
        fm = s.clone()                                           # :135
        fm.data.masked_fill_(frame_mask, MIN_VALUE)              # :143-144 (pre-softmax)
    a = torch.softmax(s, dim=1)                                  # :146
    src = ctx if ctx is not None else p_ctx
    out = torch.bmm(a.unsqueeze(1), src).squeeze(1)              # :149-154
    return out, a, fm


def dot_attention(h: Tensor, p_ctx: Tensor, ctx: Optional[Tensor], mask: Optional[Tensor],
                  frame_mask: Optional[Tensor], w_h: Tensor, b_h: Tensor, temp: float = 1.0,
                  with_sentinel: bool = False) -> Tuple[Tensor, Tensor, Optional[Tensor]]:
    """model/modules.py:24-76 (SoftAttention.forward).  with_sentinel: -inf fill (:40-41, :53-55) instead of -1e8."""
    MIN_VALUE = -float("inf") if with_sentinel else globals()["MIN_VALUE"]
    q = F.linear(h, w_h, b_h)                                    # :31
    s = torch.bmm(p_ctx, q.unsqueeze(2)).squeeze(2)              # :34-35
    s = s / temp                                                 # :37
    if mask is not None:
        s.data.masked_fill_(mask, MIN_VALUE)                     # :46
    fm = None
    if frame_mask is not None:
        fm = s.clone()                                           # :52
        fm.data.masked_fill_(frame_mask, MIN_VALUE)              # :60-61
    a = torch.softmax(s, dim=1)                                  # :63
    src = ctx if ctx is not None else p_ctx
    out = torch.bmm(a.unsqueeze(1), src).squeeze(1)              # :66-71
    return out, a, fm


# ----------------------------------------------------------------------------- LSTM cell
def lstm_cell(x: Tensor, h: Tensor, c: Tensor, w_ih: Tensor, w_hh: Tensor, b_ih: Tensor, b_hh: Tensor
              ) -> Tuple[Tensor, Tensor]:
    """nn.LSTMCell as used at model/decoder_core.py:14,27,50,61 (gate order i,f,g,o)."""
    return torch._VF.lstm_cell(x, (h, c), w_ih, w_hh, b_ih, b_hh)


def lstm_cell_explicit(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """Same cell spelled out (used to cross-check the fused ATen op in the oracle tests)."""
    g = F.linear(x, w_ih, b_ih) + F.linear(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, dim=1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    return torch.sigmoid(o) * torch.tanh(c2), c2


def gru_sequence(x: Tensor, P: Dict[str, Tensor], num_layers: int = 2, bidirectional: bool = True) -> Tensor:
    """nn.GRU(batch_first=True, h0 = 0) in eval mode, spelled out step by step: the frame-context encoder built at
    model/backbone.py:103-106 and run at :335-338.  P holds the module's own parameter names (weight_ih_l0,
    weight_hh_l0_reverse, ...); gate order r, z, n; n = tanh(W_in x + b_in + r * (W_hn h + b_hn)); h' = (1-z) n + z h."""
    cur = x
    for l in range(num_layers):
        outs = []
        for sfx in ([""] + (["_reverse"] if bidirectional else [])):
            w_ih, w_hh = P[f"weight_ih_l{l}{sfx}"], P[f"weight_hh_l{l}{sfx}"]
            b_ih, b_hh = P[f"bias_ih_l{l}{sfx}"], P[f"bias_hh_l{l}{sfx}"]
            H = w_hh.shape[1]
            h = cur.new_zeros(cur.shape[0], H)
            gi_all = F.linear(cur, w_ih, b_ih)
            ys = [None] * cur.shape[1]
            order = range(cur.shape[1]) if sfx == "" else range(cur.shape[1] - 1, -1, -1)
            for t in order:
                gh = F.linear(h, w_hh, b_hh)
                i_r, i_z, i_n = gi_all[:, t].chunk(3, 1)
                h_r, h_z, h_n = gh.chunk(3, 1)
                r, z = torch.sigmoid(i_r + h_r), torch.sigmoid(i_z + h_z)
                n = torch.tanh(i_n + r * h_n)
                h = (1 - z) * n + z * h
                ys[t] = h
            outs.append(torch.stack(ys, 1))
        cur = torch.cat(outs, 2)
    return cur


def _cell(P: Dict[str, Tensor], prefix: str):
    return (P[prefix + ".weight_ih"], P[prefix + ".weight_hh"], P[prefix + ".bias_ih"], P[prefix + ".bias_hh"])


# ----------------------------------------------------------------------------- embed / logits
def embed(P: Dict[str, Tensor], words: Tensor, drop_mask: Optional[Tensor] = None) -> Tensor:
    """model/captioner.py:53-68: Embedding -> ReLU -> Dropout (identity in eval; an explicit
    keep-mask/(1-p) can be injected for train-mode parity)."""
    e = torch.relu(F.embedding(words, P["embed.0.weight"]))
    return e if drop_mask is None else e * drop_mask


def logits_logsoftmax(P: Dict[str, Tensor], out: Tensor) -> Tensor:
    """model/captioner.py:266,361,437: F.log_softmax(self.logit(output), dim=1)."""
    return F.log_softmax(F.linear(out, P["logit.weight"], P["logit.bias"]), dim=1)


# ----------------------------------------------------------------------------- decoder steps
def decoder_step(P: Dict[str, Tensor], emb: Tensor, fc: Tensor, conv: Tensor, p_conv: Tensor, pool: Tensor,
                 p_pool: Tensor, mask: Optional[Tensor], state: Tuple[Tensor, Tensor],
                 frame_mask: Optional[Tensor] = None, *, global_img: bool = True, softattn_type: str = "additive",
                 temp: float = 1.0, out_drop: Optional[Tensor] = None, prefix: str = "decoder_core"):
    """model/decoder_core.py:30-66 (TopDownDecoderCore.forward)."""
    h, c = state
    prev_h = h[-1]                                                           # :43 (language LSTM's h)
    x_att = torch.cat([prev_h, fc, emb], 1) if global_img else torch.cat([prev_h, emb], 1)  # :45-48
    h_att, c_att = lstm_cell(x_att, h[0], c[0], *_cell(P, prefix + ".att_lstm"))            # :50
    wh, bh = P[prefix + ".soft_attn.h2attn.weight"], P[prefix + ".soft_attn.h2attn.bias"]
    if softattn_type == "additive":
        wa, ba = P[prefix + ".soft_attn.alpha_net.weight"], P[prefix + ".soft_attn.alpha_net.bias"]
        ctx_r, a_r, fm = additive_attention(h_att, p_pool, pool, mask, frame_mask, wh, bh, wa, ba)  # :54-55
        ctx_f, _, _ = additive_attention(h_att, p_conv, conv, None, None, wh, bh, wa, ba)           # :56
    else:
        ctx_r, a_r, fm = dot_attention(h_att, p_pool, pool, mask, frame_mask, wh, bh, temp)
        ctx_f, _, _ = dot_attention(h_att, p_conv, conv, None, None, wh, bh, temp)
    x_lang = torch.cat([ctx_r + ctx_f, h_att], 1)                                            # :59
    h_lang, c_lang = lstm_cell(x_lang, h[1], c[1], *_cell(P, prefix + ".lang_lstm"))         # :61
    out = h_lang if out_drop is None else h_lang * out_drop                                  # :62
    return out, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang])), a_r, fm, ctx_r  # :64-66


def reconstructor_step(P: Dict[str, Tensor], emb: Tensor, fc: Tensor, loc_pool: Tensor, loc_conv: Tensor,
                       state: Tuple[Tensor, Tensor], *, global_img: bool = True,
                       out_drop: Optional[Tensor] = None):
    """model/decoder_core.py:86-113 (AttenedDecoderCore.forward); shares the decoder's LSTM
    cells (model/captioner.py:86-87)."""
    h, c = state
    prev_h = h[-1]
    x_att = torch.cat([prev_h, fc, emb], 1) if global_img else torch.cat([prev_h, emb], 1)   # :99-102
    h_att, c_att = lstm_cell(x_att, h[0], c[0], *_cell(P, "decoder_core.att_lstm"))          # :104
    x_lang = torch.cat([loc_pool + loc_conv, h_att], 1)                                      # :106
    h_lang, c_lang = lstm_cell(x_lang, h[1], c[1], *_cell(P, "decoder_core.lang_lstm"))      # :108
    out = h_lang if out_drop is None else h_lang * out_drop
    return out, (torch.stack([h_att, h_lang]), torch.stack([c_att, c_lang]))                 # :111-113


def localizer_step(P: Dict[str, Tensor], emb: Tensor, conv: Tensor, p_conv: Tensor, pool: Tensor, p_pool: Tensor,
                   mask: Optional[Tensor], frame_mask: Optional[Tensor] = None, *, temp: float = 1.0):
    """model/localizer_core.py:17-40 (LocalizerNoLSTMCore.forward): dot-product attention with
    the embedded word as query; state passes through untouched."""
    wh, bh = P["localizer_core.soft_attn.h2attn.weight"], P["localizer_core.soft_attn.h2attn.bias"]
    loc_pool, prob, _ = dot_attention(emb, p_pool, pool, mask, frame_mask, wh, bh, temp)     # :36-37
    loc_conv, _, _ = dot_attention(emb, p_conv, conv, None, None, wh, bh, temp)              # :39
    return loc_pool, loc_conv, prob


def init_hidden(batch: int, R: int, layers: int = 2):
    """model/captioner.py:96-101."""
    return torch.zeros(layers, batch, R), torch.zeros(layers, batch, R)


# ----------------------------------------------------------------------------- greedy sampler
def greedy_sample(P: Dict[str, Tensor], feats: Dict[str, Tensor], T: int, unk_idx: int, *,
                  global_img: bool = True, softattn_type: str = "additive", temp: float = 1.0,
                  return_logprobs: bool = False):
    """model/captioner.py:384-443 (_sample): exactly T decoder steps from BOS=0, never stops at
    EOS; top-2 with UNK suppression (:415-422); returns post-softmax region attention."""
    B = feats["fc_feats"].size(0)
    state = init_hidden(B, feats["fc_feats"].size(1))
    mask = feats["pnt_mask"][:, 1:]
    seq, atts, lps, all_logp = [], [], [], []
    logp = None
    for t in range(T + 1):
        if t == 0:
            word = torch.zeros(B, dtype=torch.long)                         # :413
        else:
            top_lp, top_w = torch.topk(logp.data, 2, dim=1)                 # :415-416
            not_unk = top_w[:, 0] != unk_idx                                # :417
            lp = torch.where(not_unk, top_lp[:, 0], top_lp[:, 1])           # :418-419
            word = torch.where(not_unk, top_w[:, 0], top_w[:, 1]).long()    # :420-422
            seq.append(word)
            lps.append(lp)
        if t < T:
            e = embed(P, word)                                              # :424
            out, state, a_r, _, _ = decoder_step(P, e, feats["fc_feats"], feats["conv_feats"],
                                                 feats["p_conv_feats"], feats["pool_feats"],
                                                 feats["p_pool_feats"], mask, state, None,
                                                 global_img=global_img, softattn_type=softattn_type, temp=temp)
            logp = logits_logsoftmax(P, out)                                # :437
            atts.append(a_r)
            all_logp.append(logp)
    res = (torch.stack(seq, 1), torch.stack(atts, 1))                       # :440-443
    if return_logprobs:
        res = res + (torch.stack(lps, 1), torch.stack(all_logp, 1))
    return res


# ----------------------------------------------------------------------------- beam search (build-defined)
def beam_search(P: Dict[str, Tensor], feats: Dict[str, Tensor], T: int, unk_idx: int, beam: int, *,
                global_img: bool = True, softattn_type: str = "additive"):
    """Build-defined (the reference asserts beam_size == 1, trainer.py:218): SURVEY.md section 7
    "Beam-search specification".  Cumulative log-prob, no length normalisation; step 0 expands the
    single BOS hypothesis; logp[unk] = -inf; a hypothesis that emitted 0 is frozen (carries its
    score, emits 0); always T steps; ties -> lower (parent, word) flat index.
    Returns rank-0 seq [B,T], its att2_weights [B,T,N], and all beam scores [B,beam]."""
    B, R = feats["fc_feats"].shape
    N = feats["pool_feats"].size(1)
    V = P["logit.weight"].size(0)
    mask = feats["pnt_mask"][:, 1:]
    rep = lambda x: x.repeat_interleave(beam, 0)
    fc, conv, pconv, pool, ppool = [rep(feats[k]) for k in
                                    ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats")]
    m = rep(mask)
    state = init_hidden(B * beam, R)
    words = torch.zeros(B * beam, dtype=torch.long)
    scores = torch.zeros(B, beam)
    scores[:, 1:] = -float("inf")            # step 0: one live hypothesis per clip
    done = torch.zeros(B, beam, dtype=torch.bool)
    seqs = torch.zeros(B, beam, 0, dtype=torch.long)
    atts = torch.zeros(B, beam, 0, N)
    for t in range(T):
        e = embed(P, words)
        out, state, a_r, _, _ = decoder_step(P, e, fc, conv, pconv, pool, ppool, m, state, None,
                                             global_img=global_img, softattn_type=softattn_type)
        logp = logits_logsoftmax(P, out).view(B, beam, V).clone()
        logp[:, :, unk_idx] = -float("inf")
        frozen = torch.full((V,), -float("inf"))
        frozen[0] = 0.0
        logp = torch.where(done.unsqueeze(2), frozen.view(1, 1, V), logp)
        cand = (scores.unsqueeze(2) + logp).view(B, beam * V)
        # stable selection: sort by (-score, flat index)
        order = torch.argsort(-cand, dim=1, stable=True)[:, :beam]
        scores = torch.gather(cand, 1, order)
        parent, word = order // V, order % V
        gidx = (parent + torch.arange(B).unsqueeze(1) * beam).view(-1)
        state = (state[0][:, gidx], state[1][:, gidx])
        seqs = torch.cat([torch.gather(seqs, 1, parent.unsqueeze(2).expand(-1, -1, seqs.size(2))),
                          word.unsqueeze(2)], 2)
        a_sel = a_r.view(B, beam, N)
        a_sel = torch.gather(a_sel, 1, parent.unsqueeze(2).expand(-1, -1, N))
        atts = torch.cat([torch.gather(atts, 1, parent.view(B, beam, 1, 1).expand(-1, -1, atts.size(2), N)),
                          a_sel.unsqueeze(2)], 2)
        done = torch.gather(done, 1, parent) | (word == 0)
        words = word.view(-1)
    return seqs[:, 0], atts[:, 0], scores


def exhaustive_best_sequence(P, feats, T: int, unk_idx: int):
    """Brute force over all V^T sequences (tiny V, T only) under the beam-search scoring rule:
    returns the best total log-prob per clip -- an upper bound any beam width must respect and
    that beam >= V^(T-1) must attain."""
    B, R = feats["fc_feats"].shape
    V = P["logit.weight"].size(0)
    mask = feats["pnt_mask"][:, 1:]
    best = torch.full((B,), -float("inf"))
    for tail in itertools.product(range(V), repeat=T):
        if unk_idx in tail:
            continue
        state = init_hidden(B, R)
        word = torch.zeros(B, dtype=torch.long)
        tot = torch.zeros(B)
        ended = False
        ok = True
        for t in range(T):
            e = embed(P, word)
            out, state, _, _, _ = decoder_step(P, e, feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"],
                                               feats["pool_feats"], feats["p_pool_feats"], mask, state, None)
            logp = logits_logsoftmax(P, out)
            w = tail[t]
            if ended:
                if w != 0:
                    ok = False
                    break
            else:
                tot = tot + logp[:, w]
            if w == 0:
                ended = True
            word = torch.full((B,), w, dtype=torch.long)
        if ok:
            best = torch.maximum(best, tot)
    return best


# ----------------------------------------------------------------------------- grounder / criteria
def grounder(xt: Tensor, att_feats: Tensor, mask: Tensor, bias: Optional[Tensor] = None) -> Tensor:
    """model/captioner.py:132-173 (_grounder), dot-product branch (:154-158; the captioner has no
    alpha_net)."""
    dot = torch.matmul(xt, att_feats.permute(0, 2, 1).contiguous())          # :157-158
    if bias is not None:
        dot = dot + bias                                                      # :160-162
    m = mask.unsqueeze(1).expand_as(dot) if mask.dim() == 2 else mask         # :164-167
    return dot.masked_fill(m, MIN_VALUE)                                      # :171


def _text_mask(target: Tensor) -> Tensor:
    m = target.gt(0)
    return torch.cat([torch.ones_like(m[:, :1]), m[:, :-1]], 1)               # misc/utils.py:135-137


def language_criterion(logp: Tensor, target: Tensor) -> Tensor:
    """misc/utils.py:181-192 (LanguageCriterion.forward); logp is [B*T, V]."""
    m = _text_mask(target)
    sel = torch.gather(logp, 1, target.reshape(-1, 1))
    return (-sel.masked_select(m.reshape(-1, 1))).mean()


def lm_criterion(logp: Tensor, att2_w: Tensor, ground_w: Tensor, target: Tensor, att2_target: Tensor,
                 vocab_size: int):
    """misc/utils.py:132-172 (LMCriterion.forward)."""
    assert int((target >= vocab_size).sum()) == 0                              # :134
    loss = language_criterion(logp, target)                                    # :135-146
    if int(att2_target.sum()) != 0:                                            # :150
        att2 = -F.log_softmax(att2_w, dim=2).masked_select(att2_target).mean()     # :152-154
        ground = -F.log_softmax(ground_w, dim=2).masked_select(att2_target).mean()  # :157-159
    else:
        att2, ground = torch.zeros(1), torch.zeros(1)                          # :161-162
    return loss, att2, ground


# ----------------------------------------------------------------------------- label glue
def bbox_overlaps(rois: Tensor, gt: Tensor, frm_mask: Tensor) -> Tensor:
    """misc/utils.py:335-338 -> misc/bbox_transform.py:224-272 (3-d anchors branch):
    IoU with +1 pixel convention, zeroed where proposal/GT frames differ, 0 for degenerate GT
    boxes, -1 for degenerate proposals."""
    a, g = rois[:, :, :4], gt[:, :, :4]
    gx, gy = g[:, :, 2] - g[:, :, 0] + 1, g[:, :, 3] - g[:, :, 1] + 1
    ax, ay = a[:, :, 2] - a[:, :, 0] + 1, a[:, :, 3] - a[:, :, 1] + 1
    g_area, a_area = (gx * gy).unsqueeze(1), (ax * ay).unsqueeze(2)
    iw = (torch.min(a[:, :, None, 2], g[:, None, :, 2]) - torch.max(a[:, :, None, 0], g[:, None, :, 0]) + 1).clamp(min=0)
    ih = (torch.min(a[:, :, None, 3], g[:, None, :, 3]) - torch.max(a[:, :, None, 1], g[:, None, :, 1]) + 1).clamp(min=0)
    ov = iw * ih / (a_area + g_area - iw * ih)
    ov = ov * (~frm_mask).to(ov.dtype)
    ov = ov.masked_fill(((gx == 1) & (gy == 1)).unsqueeze(1).expand_as(ov), 0)
    ov = ov.masked_fill(((ax == 1) & (ay == 1)).unsqueeze(2).expand_as(ov), -1)
    return ov


def bbox_target(box_mask_t: Tensor, overlaps: Tensor) -> Tensor:
    """misc/utils.py:351-373 (bbox_target) -- the returned labels only (the seq_update side effect
    is dead: it writes a clone nobody reads, captioner.py:217,246-247)."""
    ov = overlaps.masked_fill(box_mask_t.reshape(overlaps.size(0), 1, -1).expand_as(overlaps), 0)
    return ov.max(2)[0] > 0.5


def frame_mask_on_proposals(box_mask_t: Tensor, frm_mask: Tensor, pnt_mask: Tensor) -> Tensor:
    """model/captioner.py:250-260: proposal n is masked for word t unless it shares a frame with
    a GT box that grounds word t; column 0 is the sentinel."""
    B, N, K = frm_mask.shape
    bm = box_mask_t.reshape(B, 1, K).expand(B, N, K)
    on_prop = (~(bm | frm_mask)).sum(2) <= 0
    return torch.cat([torch.zeros(B, 1, dtype=torch.bool), on_prop], 1) | pnt_mask.bool()


# ----------------------------------------------------------------------------- cyclical forward
def cyclical_forward(P: Dict[str, Tensor], feats: Dict[str, Tensor], batch: Dict[str, Tensor], *, T: int,
                     vocab_size: int, train_decoder_only: bool = False, global_img: bool = True,
                     softattn_type: str = "additive", localizer_temp: float = 1.0, seq_per_img: int = 1,
                     collect: Optional[dict] = None, dropout: Optional[Dict[str, Tensor]] = None):
    """model/captioner.py:196-382 (_forward_3_loops).  feats is what the once-per-clip encoder returned (:231-233); returns
    (lm, att2, ground, cls[, lm_recon]).  dropout = None: eval mode.  Train mode (model.train(): nn.Dropout on the embedded word,
    captioner.py:53-68, and on the language LSTM's output, decoder_core.py:62, 109) with DICTATED masks -- the reference draws
    them from the global generator; for parity with another implementation they are inputs here: keep / (1 - p) tensors
    "emb_a" / "emb_b" / "emb_c" [B, T, E] (loops A, B, C), "out_a" / "out_c" [T, B, R], and "vis_embed" [B, T, G] for the grounder's
    class embeddings (roi_feat_extractor.vis_embed is Embedding -> ReLU -> Dropout, model/backbone.py:55-57)."""
    dm = (lambda key, t: None) if dropout is None else (lambda key, t: dropout[key][:, t] if key.startswith("emb") else dropout[key][t])
    fc, conv, pconv = feats["fc_feats"], feats["conv_feats"], feats["p_conv_feats"]
    pool, ppool, gpool, pnt_mask = feats["pool_feats"], feats["p_pool_feats"], feats["g_pool_feats"], feats["pnt_mask"]
    B, N = pool.size(0), pool.size(1)
    gt = batch["gt_seq"][:, :seq_per_img, :].reshape(-1, batch["gt_seq"].size(2))            # :210-211
    gt = torch.cat([torch.zeros(gt.size(0), 1, dtype=gt.dtype), gt], 1)                       # :212-213
    iseq = batch["input_seq"].reshape(-1, batch["input_seq"].size(2), batch["input_seq"].size(3))  # :216
    box_mask, frm_mask = batch["box_mask"], batch["frm_mask"]
    overlaps = bbox_overlaps(batch["proposals"], batch["gt_bboxs"],
                             frm_mask | pnt_mask[:, 1:].unsqueeze(-1))                        # :228-229
    state = init_hidden(B, fc.size(1))
    mask = pnt_mask[:, 1:]
    logps, fmasked, roi_labels, fmo = [], [], [], []
    for t in range(T):                                                                        # :242 Loop A
        e = embed(P, gt[:, t], dm("emb_a", t))                                                # :243-244
        roi_labels.append(bbox_target(box_mask[:, :, :, t + 1], overlaps))                    # :246-248
        fm_t = frame_mask_on_proposals(box_mask[:, 0, :, t + 1], frm_mask, pnt_mask)          # :251-260
        fmo.append(fm_t)
        out, state, a_r, fm, _ = decoder_step(P, e, fc, conv, pconv, pool, ppool, mask, state, fm_t[:, 1:],
                                              global_img=global_img, softattn_type=softattn_type,
                                              out_drop=dm("out_a", t))                           # :262-264
        logps.append(logits_logsoftmax(P, out))                                               # :266
        fmasked.append(fm)
    att2_w = torch.stack(fmasked, 1)                                                          # :273 (pre-softmax)
    lang = torch.stack(logps, 1)                                                              # :276
    roi_labels = torch.stack(roi_labels, 1)                                                   # :277
    fmo = torch.stack(fmo, 1)                                                                 # :278-279
    xt_clamp = torch.clamp(iseq[:, 1:T + 1, 0] - vocab_size, min=0)                           # :282-283
    xt_all = torch.relu(F.embedding(xt_clamp, P["roi_feat_extractor.vis_embed.0.weight"]))   # :284
    if dropout is not None:
        xt_all = xt_all * dropout["vis_embed"]                                                # backbone.py:57 in train mode
    bias = P["roi_feat_extractor.vis_classifiers_bias"][xt_clamp].unsqueeze(2).expand(B, T, N)  # :287-288
    ground_w = grounder(xt_all, gpool, fmo[:, :, 1:], bias + att2_w)                          # :293-294
    target = gt[:, 1:T + 1]
    cls_loss = feats.get("cls_loss", torch.zeros(()))
    lm, att2, ground = lm_criterion(lang.view(-1, lang.size(2)), att2_w, ground_w, target,
                                    roi_labels[:, :T, :], vocab_size)                         # :300-305 / :368-373
    if collect is not None:
        collect.update(lang=lang, att2_weights=att2_w, ground_weights=ground_w, roi_labels=roi_labels,
                       frm_mask_output=fmo, overlaps=overlaps)
    if train_decoder_only:                                                                    # :297-307
        return lm.reshape(1), att2.reshape(1), ground.reshape(1), cls_loss.reshape(1)
    out_seq = lang.max(2)[1]                                                                  # :313 argmax cut
    loc_pool, loc_conv = [], []
    for t in range(T):                                                                        # :320 Loop B
        e = embed(P, out_seq[:, t], dm("emb_b", t))
        lp_, lc_, _ = localizer_step(P, e, conv, pconv, pool, ppool, mask, fmo[:, t, 1:], temp=localizer_temp)
        loc_pool.append(lp_)
        loc_conv.append(lc_)
    state = init_hidden(B, fc.size(1))                                                        # :346-347
    rec = []
    for t in range(T):                                                                        # :348 Loop C
        e = embed(P, gt[:, t], dm("emb_c", t))
        out, state = reconstructor_step(P, e, fc, loc_pool[t], loc_conv[t], state, global_img=global_img,
                                        out_drop=dm("out_c", t))
        rec.append(logits_logsoftmax(P, out))                                                 # :361
    rec = torch.stack(rec, 1)
    lm_recon = language_criterion(rec.view(-1, rec.size(2)), target)                          # :378-379
    if collect is not None:
        collect.update(output_seq=out_seq, consistent=rec, loc_pool=torch.stack(loc_pool, 1),
                       loc_conv=torch.stack(loc_conv, 1))
    return lm.reshape(1), att2.reshape(1), ground.reshape(1), cls_loss.reshape(1), lm_recon.reshape(1)


def training_loss(losses: Sequence[Tensor], *, xe_loss_weight: float, w_att2: float, w_cls: float,
                  caption_consistency_loss_weight: float) -> Tensor:
    """trainer.py:101-109: ground_loss is returned by the model but never optimised."""
    lm, att2, _ground, cls = losses[:4]
    loss = xe_loss_weight * lm.mean() + w_att2 * att2.mean() + w_cls * cls.mean()
    if len(losses) > 4:
        loss = loss + caption_consistency_loss_weight * losses[4].mean()
    return loss


def to_torch(d):
    """numpy dict -> torch dict (float32 / int64 / bool preserved)."""
    import numpy as np
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}
