"""Criteria, label glue and small helpers -- the parts of the reference's misc/utils.py the
caption hot path touches: LMCriterion / LanguageCriterion
(misc/utils.py:127-192), bbox_overlaps / bbox_target (:335-373), update_values (:58-63),
decode_sequence (:100-116), AverageMeter (:501-517).  The dead NBT image utilities are not
reproduced.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import functional as F_


def update_values(dict_from, dict_to):
    """YAML overlay: non-None values of dict_from win, recursively (reference :58-63)."""
    for key, value in dict_from.items():
        if isinstance(value, dict):
            update_values(dict_from[key], dict_to[key])
        elif value is not None:
            dict_to[key] = dict_from[key]


def decode_sequence(itow, itod, ltow, itoc, wtod, seq, vocab_size, opt):
    """Word indices -> sentences, stopping at the first 0 (reference :100-116; the unused
    detection-vocabulary arguments are kept for signature compatibility)."""
    out = []
    for row in seq.tolist():
        words = []
        for ix in row:
            if ix == 0:
                break
            words.append(itow[str(ix)])
        # the reference emits a separating blank before it sees the terminating 0
        txt = ' '.join(words)
        if len(words) < len(row) and len(words) > 0:
            txt += ' '
        out.append(txt)
    return out


def _text_mask(target):
    m = target.gt(0)
    return torch.cat([torch.ones_like(m[:, :1]), m[:, :-1]], 1)   # includes the first EOS (:135-137)


def _masked_nll_mean(txt_input, target):
    mask = _text_mask(target)
    w = mask.reshape(-1).to(torch.float32)
    total = F_.masked_nll_sum(txt_input, target.reshape(-1), w)
    count = w.sum()
    return (total / count).reshape(())


def _masked_nll_mean_from_logits(logits, target, t_major=False):
    """Same value as _masked_nll_mean(log_softmax(logits), target), plus the per-row argmax, from one fused pass.
    t_major: the rows of `logits` are ordered t * B + b (the C-driven training loops' outputs) instead of b * T + t."""
    w = _text_mask(target).to(torch.float32)
    if t_major:
        total, argmax = F_.vocab_nll(logits, target.t().reshape(-1), w.t().reshape(-1))
        return (total / w.sum()).reshape(()), argmax.view(target.shape[1], target.shape[0]).t()
    total, argmax = F_.vocab_nll(logits, target.reshape(-1), w.reshape(-1))
    return (total / w.sum()).reshape(()), argmax.view(target.shape)


def _masked_nll_mean_from_head(x, weight, bias, target, t_major=False, done=None, compute_only=False):
    """_masked_nll_mean_from_logits(x W^T + b, target) with the criterion folded into the vocabulary head's GEMM finish: the logits
    are never materialised (cvc.functional.vocab_head_nll); falls back to the two-step form for shapes that kernel does not take."""
    if not F_.vocab_head_nll_ok(x, weight):
        return _masked_nll_mean_from_logits(F_.linear(x, weight, bias), target, t_major)
    # the loss mask, its count and the t-major copies depend on the target alone: both heads of a cyclical pass (decode,
    # reconstruction) score against the same target tensor, so they are formed once and kept on it
    pre = getattr(target, "_cvc_nll_pre", None)
    if pre is None or pre[0] != (target._version, t_major):
        w = _text_mask(target).to(torch.float32)
        tf, wf = (target.t().reshape(-1), w.t().reshape(-1)) if t_major else (target.reshape(-1), w.reshape(-1))
        pre = ((target._version, t_major), tf.contiguous(), wf.contiguous(), w.sum())
        target._cvc_nll_pre = pre
    _, tf, wf, count = pre
    if compute_only:       # -> (opaque result for a later call with done=, the argmax words)
        done = F_.vocab_head_nll_compute(x, weight, bias, tf, wf)
        argmax = done[1]
        return done, (argmax.view(target.shape[1], target.shape[0]).t() if t_major else argmax.view(target.shape))
    total, argmax = F_.vocab_head_nll(x, weight, bias, tf, wf, done)
    if t_major:
        return (total / count).reshape(()), argmax.view(target.shape[1], target.shape[0]).t()
    return (total / count).reshape(()), argmax.view(target.shape)


class LMCriterion(nn.Module):
    """reference misc/utils.py:127-172"""

    def __init__(self, opt):
        super().__init__()
        self.vocab_size = opt.vocab_size

    def forward(self, txt_input, att2_weights, ground_weights, target, att2_target, input_seq):
        if not torch.cuda.is_current_stream_capturing():
            assert torch.sum(target >= self.vocab_size) == 0                  # reference :134 (host check; skipped under capture)
        loss = _masked_nll_mean(txt_input, target)
        return (loss, *self.attention_losses(att2_weights, ground_weights, att2_target))

    def from_logits(self, logits, att2_weights, ground_weights, target, att2_target, input_seq, t_major=False):
        """forward() fed raw logits instead of log-probs: -> (lm_loss, att2_loss, ground_loss, argmax [B, T]).
        SURVEY section 8(f) rank 2: the criterion folded into the vocabulary head's epilogue pass."""
        if not torch.cuda.is_current_stream_capturing():
            assert torch.sum(target >= self.vocab_size) == 0
        loss, argmax = _masked_nll_mean_from_logits(logits, target, t_major)
        return (loss, *self.attention_losses(att2_weights, ground_weights, att2_target), argmax)

    def from_head(self, x, head, att2_weights, ground_weights, target, att2_target, input_seq, t_major=False, done=None):
        """from_logits() fed the vocabulary head's INPUT x [rows, R] and the head module (nn.Linear): the head's GEMM and the
        criterion run as one op, no logits tensor.  done: head_words()'s first result for the same x (the arithmetic then is not
        repeated)."""
        if not torch.cuda.is_current_stream_capturing():
            assert torch.sum(target >= self.vocab_size) == 0
        loss, argmax = _masked_nll_mean_from_head(x, head.weight, head.bias, target, t_major, done=done)
        return (loss, *self.attention_losses(att2_weights, ground_weights, att2_target), argmax)

    @staticmethod
    def head_words(x, head, target, t_major=False):
        """the head + criterion arithmetic of from_head() ahead of its graph node: -> (done, argmax words [B, T]), or None for
        shapes the fused head does not take"""
        if not F_.vocab_head_nll_ok(x, head.weight):
            return None
        return _masked_nll_mean_from_head(x, head.weight, head.bias, target, t_major, compute_only=True)

    @staticmethod
    def attention_losses(att2_weights, ground_weights, att2_target):
        # supervised attention / grounding losses (w_att2 = 0 by default).  The reference
        # branches on `att2_target.sum() != 0` and uses masked_select (:150-162); the same value without a host round
        # trip or a data-dependent shape: -sum(log_softmax * target) / max(count, 1), which is 0 when nothing is labelled.
        if att2_weights.is_cuda and att2_weights.dtype == torch.float32 and ground_weights.dtype == torch.float32 \
                and att2_weights.dim() == 3 and att2_weights.stride(2) == 1 and ground_weights.stride(2) == 1:
            # both criteria of all T steps from one kernel pair (csrc/label_glue.hip), no library log_softmax / masked sums
            return F_.attn_nll(att2_weights, ground_weights, att2_target)
        tgt = att2_target.to(att2_weights.dtype)
        count = tgt.sum().clamp(min=1.0)
        att2_loss = (-(F.log_softmax(att2_weights, dim=2) * tgt).sum() / count).reshape(1)
        ground_loss = (-(F.log_softmax(ground_weights, dim=2) * tgt).sum() / count).reshape(1)
        return att2_loss, ground_loss


class LanguageCriterion(nn.Module):
    """reference misc/utils.py:175-192"""

    def forward(self, txt_input, target):
        return _masked_nll_mean(txt_input, target)

    def from_logits(self, logits, target, t_major=False):
        return _masked_nll_mean_from_logits(logits, target, t_major)[0]

    def from_head(self, x, head, target, t_major=False):
        return _masked_nll_mean_from_head(x, head.weight, head.bias, target, t_major)[0]


def bbox_overlaps(rois, gt_box, frm_mask):
    """IoU of proposals [B,N,>=5] vs GT boxes [B,K,>=5] (reference :335-338 ->
    misc/bbox_transform.py:224-272): +1-pixel convention, zero where frm_mask, 0 for degenerate GT
    boxes, -1 for degenerate proposals."""
    a, g = rois[:, :, :4], gt_box[:, :, :4]
    gx, gy = g[:, :, 2] - g[:, :, 0] + 1, g[:, :, 3] - g[:, :, 1] + 1
    ax, ay = a[:, :, 2] - a[:, :, 0] + 1, a[:, :, 3] - a[:, :, 1] + 1
    iw = (torch.min(a[:, :, None, 2], g[:, None, :, 2]) - torch.max(a[:, :, None, 0], g[:, None, :, 0]) + 1).clamp(min=0)
    ih = (torch.min(a[:, :, None, 3], g[:, None, :, 3]) - torch.max(a[:, :, None, 1], g[:, None, :, 1]) + 1).clamp(min=0)
    inter = iw * ih
    ov = inter / ((ax * ay).unsqueeze(2) + (gx * gy).unsqueeze(1) - inter)
    ov = ov * (~frm_mask).to(ov.dtype)
    ov = ov.masked_fill(((gx == 1) & (gy == 1)).unsqueeze(1).expand_as(ov), 0)
    ov = ov.masked_fill(((ax == 1) & (ay == 1)).unsqueeze(2).expand_as(ov), -1)
    return ov


def bbox_target(mask, overlaps, seq, seq_update, vocab_size):
    """Per-word proposal labels (reference :351-373): proposal n is positive for the word if it
    overlaps (IoU > 0.5) a GT box that grounds the word.  Also reproduces the (deprecated)
    seq_update side effect."""
    B = overlaps.size(0)
    ov = overlaps.masked_fill(mask.reshape(B, 1, -1).expand_as(overlaps), 0)
    labels = ov.max(2)[0] > 0.5
    no_proposal_idx = (labels.sum(1) > 0) != (seq[:, 2] > 0)
    # same writes as the reference's guarded block, expressed without a host-side `if .sum() > 0` (a
    # device->host sync per decode step that would keep the launch queue from running ahead)
    seq_update[:, 0] = torch.where(no_proposal_idx, seq_update[:, 3], seq_update[:, 0])
    seq_update[:, 1] = torch.where(no_proposal_idx, torch.zeros_like(seq_update[:, 1]), seq_update[:, 1])
    seq_update[:, 2] = torch.where(no_proposal_idx, torch.zeros_like(seq_update[:, 2]), seq_update[:, 2])
    return labels


class AverageMeter(object):
    """reference :501-517"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = self.avg = self.sum = self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def set_tb_logger(log_dir, exp_name, resume):
    """TensorBoard writer if tensorboardX / torch.utils.tensorboard is importable, else None
    (the reference hard-requires tensorboardX, cycle_utils.py:14-23)."""
    import os
    import shutil
    path = log_dir + '/' + exp_name
    if not resume and os.path.exists(path):
        shutil.rmtree(path, ignore_errors=True)
    try:
        from tensorboardX import SummaryWriter
    except ImportError:
        try:
            from torch.utils.tensorboard import SummaryWriter
        except ImportError:
            return None
    return SummaryWriter(log_dir=path)
