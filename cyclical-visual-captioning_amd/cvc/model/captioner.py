"""Decode-and-ground captioner -- drop-in for the reference's model/captioner.py
(`DecodeAndGroundCaptionerGVDROI`): same constructor, same 11-tensor forward, same
state_dict keys, same returns (a tuple of [1]-shaped losses in training, `(seq, att2_weights,
None)` in inference).  What differs is where the arithmetic runs:

* inference (`_sample`, reference :384-443) goes through cvc.decode.DecodeEngine: a flat list of
  HIP launches per step, no host round trips, optional HIP-graph replay; `beam_size > 1` (which
  the reference only asserts away, trainer.py:218) is implemented per SURVEY.md section 7;
* training (`_forward_3_loops`, reference :196-382) runs the same three loops through the
  autograd ops of cvc.functional, with the localizer loop (no recurrence) and both vocabulary
  projections batched over T.

The once-per-clip encoder (`roi_extractor`, reference model/backbone.py) is outside the hot path:
pass any module with the reference extractor's call/return contract (the reference class itself
works); `PrecomputedRegionFeatures` below feeds pre-extracted features.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from .. import dropout
from .. import functional as F_
from .. import hip
from ..decode import DecodeEngine, DecodeWeights
from ..misc import utils
from .decoder_core import AttenedDecoderCore, TopDownDecoderCore
from .localizer_core import LocalizerNoLSTMCore


class PrecomputedRegionFeatures(nn.Module):
    """Stand-in for the once-per-clip encoder when features are pre-extracted / pre-projected
    (the benchmark's input contract).  Owns the two encoder tensors the captioner's grounder
    reads (`vis_embed.0.weight`, `vis_classifiers_bias`, reference backbone.py:55,140) and
    returns, in the order of backbone.py:350-351, the feature dict handed in as `segs_feat`."""

    KEYS = ("fc_feats", "conv_feats", "p_conv_feats", "pool_feats", "p_pool_feats", "g_pool_feats", "pnt_mask")

    def __init__(self, detect_size: int, vis_encoding_size: int, drop_prob: float = 0.5):
        super().__init__()
        self.vis_embed = nn.Sequential(nn.Embedding(detect_size + 1, vis_encoding_size), nn.ReLU(), nn.Dropout(drop_prob))
        self.vis_classifiers_bias = nn.Parameter(torch.zeros(detect_size + 1))

    def forward(self, segs_feat, proposals=None, num=None, mask_boxes=None, region_feats=None, gt_boxes=None,
                overlaps=None, sample_idx=None, eval_obj_ground=False, replicate_feat=True):
        if not isinstance(segs_feat, dict):
            raise TypeError("PrecomputedRegionFeatures expects the feature dict as the first model input")
        f = segs_feat
        cls_loss = f.get("cls_loss", f["fc_feats"].new_zeros(()))
        return (f["fc_feats"], f["conv_feats"], f["p_conv_feats"], f["pool_feats"], f["p_pool_feats"], f["g_pool_feats"],
                f["pnt_mask"], overlaps, None, cls_loss)


class DecodeAndGroundCaptionerGVDROI(nn.Module):
    """reference model/captioner.py:16-443"""

    def __init__(self, opts, pretrained_decoder=None, embed=None, logit=None, roi_extractor=None):
        super().__init__()
        self.opts = opts
        self.vocab_size = opts.vocab_size
        self.ix_to_word = getattr(opts, "itow", None)
        if roi_extractor is None:
            raise ValueError(
                "roi_extractor is required: the once-per-clip region/frame encoder (reference model/backbone.py) is "
                "outside the MI355X hot path.  Pass the reference's RegionalFeatureExtractorGVD, or "
                "cvc.model.captioner.PrecomputedRegionFeatures for pre-extracted features.")
        self.roi_feat_extractor = roi_extractor
        self.seq_length = opts.seq_length
        self.seq_per_img = opts.seq_per_img
        self.decoder_num_layers = 2
        self.localizer_num_layers = 1
        self.rnn_size = opts.rnn_size
        self.ss_prob = 0.0
        self.iou_threshold = 0.5

        self.decoder_core = TopDownDecoderCore(opts) if pretrained_decoder is None else pretrained_decoder
        vocab_rows = opts.vocab_size + 1 if opts.embedding_vocab_plus_1 else opts.vocab_size
        if embed is None:
            embed = nn.Sequential(nn.Embedding(vocab_rows, opts.input_encoding_size), nn.ReLU(),
                                  nn.Dropout(opts.drop_prob_lm))
        self.embed = embed
        self.logit = nn.Linear(opts.rnn_size, vocab_rows) if logit is None else logit
        self.localizer_core = LocalizerNoLSTMCore(opts)
        # reconstructor shares the decoder's two LSTM cells (reference :86-87)
        self.attended_roi_decoder_core = AttenedDecoderCore(opts, self.decoder_core.att_lstm, self.decoder_core.lang_lstm)
        self.critLM = utils.LMCriterion(opts)
        self.unk_idx = int(opts.wtoi['UNK'])
        self.xe_criterion = utils.LanguageCriterion()
        self.beam_size = int(getattr(opts, "beam_size", 1))
        self.use_hip_graph = bool(getattr(opts, "hip_graph", False))
        # test hook: set to a dict to receive the training pass's intermediate tensors (ground_weights, att2_weights,
        # output_seq) that the reference computes as locals of _forward_3_loops and never returns
        self.debug_collect: Optional[dict] = None

    # ------------------------------------------------------------------ small pieces
    @property
    def device(self):
        return self.logit.weight.device

    def step_capturable(self, deferred_errors: bool = False) -> bool:
        """True when a whole training step of this model can be captured into a HIP graph (cvc.trainer.Trainer.train): the hot
        path on pre-extracted features always; with the once-per-clip encoder in front (raw features, the reference's own flow:
        trainer.py:39-150 -> model/backbone.py:298-351) when the encoder says so -- its persistent recurrence kernels report barrier
        time-outs through error words, which a captured step can only carry in deferred mode (cvc.hip.defer_errors)."""
        ext = self.roi_feat_extractor
        if isinstance(ext, PrecomputedRegionFeatures):
            return True
        cap = getattr(ext, "step_capturable", None)
        return bool(cap is not None and cap(deferred_errors=deferred_errors))

    def reports_error_words(self) -> bool:
        """does a step of this model launch kernels that report through error words (the encoder's persistent GRU)?"""
        rep = getattr(self.roi_feat_extractor, "reports_error_words", None)
        return bool(rep is not None and rep())

    def init_hidden(self, batch_size, num_layers):
        """reference :96-101"""
        z = torch.zeros(num_layers, batch_size, self.rnn_size, device=self.device)
        return (z, z.clone())

    def _init_step_state(self, batch_size):
        """init_hidden() as the unstacked (h_att, c_att, h_lang, c_lang) the cores' step() takes"""
        z = torch.zeros(4, batch_size, self.rnn_size, device=self.device)
        return tuple(z.unbind(0))

    def _embed(self, word, site=None):
        """embed = Embedding -> ReLU -> Dropout (reference :53-68); lookup+ReLU(+mask) is one kernel.  site: the dropout site's
        name (cvc/dropout.py) so that a test can dictate the mask."""
        drop_p = self.embed[2].p if (self.training and len(self.embed) > 2) else 0.0
        table = self.embed[0].weight
        drop = None
        if 0 < drop_p < 1 and site is not None and dropout.in_kernel(table):
            return F_.embed_relu(table, word, rng=(dropout.rng_state(table.device), dropout.site_id(site), float(drop_p)))
        if drop_p > 0:
            drop = dropout.keep_mask(site or "emb", (word.numel(), table.shape[1]), drop_p, table.device)
        return F_.embed_relu(table, word, drop)

    def _logprobs(self, output):
        """F.log_softmax(self.logit(output), dim=1) (reference :266, :361, :437)"""
        return F_.log_softmax(self._logits(output))

    def _logits(self, output):
        """self.logit(output); the training pass hands these to the fused criteria, which never write log-probs"""
        return F_.linear(output, self.logit.weight, self.logit.bias)

    def _grounder(self, xt, att_feats, mask, bias=None, min_value=-1e8):
        """reference :132-173, dot-product branch (the captioner owns no alpha_net)."""
        assert xt.size(-1) == att_feats.size(-1)
        B, S, _ = xt.size()
        R = att_feats.size(1)
        if mask.dim() == 2:
            mask = mask.unsqueeze(1).expand(B, S, R)
        elif mask.dim() != 3:
            raise NotImplementedError
        if bias is not None:
            assert bias.numel() == B * S * R
        return F_.grounder(xt, att_feats, bias, mask)

    # ------------------------------------------------------------------ forward
    def forward(self, segs_feat, input_seq, gt_caption, num, proposals, gt_boxes, mask_boxes, region_feats, frm_mask,
                sample_idx, pnt_mask, lang_eval=False, teacher_forcing=False):
        """reference :175-194"""
        F_.new_step()
        if lang_eval is False or teacher_forcing:
            return self._forward_3_loops(segs_feat, input_seq, proposals, gt_caption, num, mask_boxes, gt_boxes,
                                         region_feats, frm_mask, sample_idx, pnt_mask)
        return self._sample(segs_feat, input_seq, proposals, gt_caption, num, mask_boxes, gt_boxes, region_feats,
                            frm_mask, sample_idx, pnt_mask)

    def _encode(self, segs_feat, proposals, num, mask_boxes, region_feats, gt_boxes, frm_mask, sample_idx, pnt_mask):
        if proposals.is_cuda and proposals.dtype == torch.float32 and gt_boxes.dtype == torch.float32:
            overlaps = hip.bbox_overlaps(proposals.data, gt_boxes.data, frm_mask.data, pnt_mask[:, 1:].data)     # one launch, bit-exact
        else:
            overlaps = utils.bbox_overlaps(proposals.data, gt_boxes.data, (frm_mask | pnt_mask[:, 1:].unsqueeze(-1)).data)
        return overlaps, self.roi_feat_extractor(segs_feat, proposals, num, mask_boxes, region_feats, gt_boxes, overlaps,
                                                 sample_idx)

    def _forward_3_loops(self, segs_feat, input_seq, proposals, gt_caption, num, mask_boxes, gt_boxes, region_feats,
                         frm_mask, sample_idx, pnt_mask):
        """reference :196-382"""
        batch_size = proposals.size(0)
        num_rois = proposals.size(1)
        T = self.seq_length
        gt_caption = gt_caption[:, :self.seq_per_img, :].clone().view(-1, gt_caption.size(2))
        gt_caption = torch.cat((gt_caption.new_zeros(gt_caption.size(0), 1), gt_caption), 1)     # BOS = 0
        input_seq = input_seq.view(-1, input_seq.size(2), input_seq.size(3))
        B = gt_caption.size(0)
        if self.training and dropout.in_kernel(self.logit.weight):
            dropout.advance(self.device)              # this pass's masks: step word of the in-kernel generator += 1, on the device

        overlaps, (fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, g_pool_feats, pnt_mask, _ov, _cls_pred,
                   cls_loss) = self._encode(segs_feat, proposals, num, mask_boxes, region_feats, gt_boxes, frm_mask,
                                            sample_idx, pnt_mask)
        region_mask = pnt_mask[:, 1:].contiguous()

        # ---- label glue for all T steps at once (reference :246-260 does it per step; none of it depends on
        # the recurrence): per-word proposal labels and the frame mask on proposals
        tgt_steps = slice(1, T + 1)
        if overlaps.is_cuda and overlaps.dtype == torch.float32:
            # one kernel for all T steps (csrc/label_glue.hip): bool results, bit-exact; the deprecated seq_update side effect of
            # bbox_target (a clone nobody reads, misc/utils.py:363-371) is not reproduced
            roi_labels, frm_mask_output, step_fmask = hip.label_glue(overlaps, mask_boxes[:, 0, :, tgt_steps], frm_mask, pnt_mask)
            return self._forward_after_glue(gt_caption, input_seq, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats,
                                            g_pool_feats, region_mask, step_fmask, frm_mask_output, roi_labels, cls_loss)
        bm = mask_boxes[:, :, :, tgt_steps]                                           # [B, seq_per_img, K, T]
        ov = overlaps.unsqueeze(1).masked_fill(bm[:, 0].permute(0, 2, 1).unsqueeze(2).expand(B, T, num_rois, -1), 0)
        roi_labels = ov.max(3)[0] > 0.5                                                # [B, T, N]   (utils.bbox_target)
        no_prop = (roi_labels.sum(2) > 0) != (input_seq[:, tgt_steps, 2] > 0)          # deprecated seq_update side effect
        upd = input_seq.data.clone()[:, tgt_steps]
        upd[..., 0] = torch.where(no_prop, upd[..., 3], upd[..., 0])
        upd[..., 1] = torch.where(no_prop, torch.zeros_like(upd[..., 1]), upd[..., 1])
        upd[..., 2] = torch.where(no_prop, torch.zeros_like(upd[..., 2]), upd[..., 2])
        box_mask_t = bm[:, 0].permute(0, 2, 1).unsqueeze(2)                            # [B, T, 1, K]
        frm_on_prop = torch.sum(~(box_mask_t | frm_mask.unsqueeze(1)), dim=3) <= 0     # [B, T, N]
        frm_mask_output = torch.cat((frm_on_prop.new_zeros(B, T, 1), frm_on_prop), dim=2) | pnt_mask.bool().unsqueeze(1)
        step_fmask = frm_mask_output[:, :, 1:].permute(1, 0, 2).contiguous()           # [T, B, N]
        return self._forward_after_glue(gt_caption, input_seq, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats,
                                        g_pool_feats, region_mask, step_fmask, frm_mask_output, roi_labels, cls_loss)

    def _forward_after_glue(self, gt_caption, input_seq, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, g_pool_feats,
                            region_mask, step_fmask, frm_mask_output, roi_labels, cls_loss):
        """_forward_3_loops from loop A on (reference :242-382)"""
        T = self.seq_length
        B = gt_caption.size(0)
        num_rois = pool_feats.size(1)
        # ---- Loop A: teacher-forced decode (sequential: LSTM recurrence)            reference :242-270
        emb_all = self._embed(gt_caption[:, :T], "emb_a")                            # [B, T, E], one launch
        loops = self._loop_plan(B, fc_feats)
        if loops is not None:
            return self._forward_loops_driven(loops, emb_all, gt_caption, input_seq, fc_feats, conv_feats, p_conv_feats, pool_feats,
                                              p_pool_feats, g_pool_feats, region_mask, step_fmask, frm_mask_output, roi_labels, cls_loss)
        state = self._init_step_state(B)
        outputs, masked_attn = [], []
        # one unbind per tensor instead of T selects: a select's backward is a zero-filled [B, T, E] tensor plus an
        # accumulation per step, unbind's is a single stack
        emb_steps = emb_all.unbind(1)
        # the att-LSTM's inputs that do not depend on the recurrence (fc_feats, the teacher-forced embedded words) are multiplied
        # for all T steps in one dense product; the per-step launches stream only the recurrent columns (cvc.functional)
        R = self.rnn_size
        hoist = torch.is_grad_enabled() and fc_feats.is_cuda
        w_att = self.decoder_core.att_lstm.weight_ih
        att_segs = (lambda e: [(fc_feats, R), (e, 2 * R)]) if self.opts.global_img_in_attn_lstm else (lambda e: [(e, R)])
        g_att = F_.hoisted_gates(w_att, att_segs(emb_all), T, B) if hoist else None
        for t in range(T):
            output, state, _roi_attn, frame_masked_attn, _wp = self.decoder_core.step(
                emb_steps[t], fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, region_mask, state,
                proposal_frame_mask=step_fmask[t], drop_site="out_a.%d" % t, gate_pre_att=None if g_att is None else g_att[t])
            outputs.append(output)
            masked_attn.append(frame_masked_attn)
        att2_weights = torch.stack(masked_attn, dim=1)                               # pre-softmax (:273)
        lang_logits = self._logits(torch.stack(outputs, 1).view(B * T, -1))          # all T at once, [B*T, V]

        # ---- grounder over all T                                                      reference :282-294
        xt_clamp = torch.clamp(input_seq[:, 1:T + 1, 0].clone() - self.vocab_size, min=0)
        xt_all = self._vis_embed(xt_clamp)
        if hasattr(self.roi_feat_extractor, 'vis_classifiers_bias'):
            bias = self.roi_feat_extractor.vis_classifiers_bias[xt_clamp].type(xt_all.type()).unsqueeze(2).expand(
                B, T, num_rois)
        else:
            bias = 0
        ground_weights = self._grounder(xt_all, g_pool_feats, frm_mask_output[:, :, 1:], bias + att2_weights)

        if self.debug_collect is not None:
            self.debug_collect.update(ground_weights=ground_weights, att2_weights=att2_weights, roi_labels=roi_labels,
                                      frm_mask_output=frm_mask_output)
        target = gt_caption[:, 1:T + 1].clone()
        # criteria fused with log_softmax and the argmax cut of :313 (one pass over the logits, no [B,T,V] log-probs)
        lm_loss, att2_loss, ground_loss, output_seq = self.critLM.from_logits(
            lang_logits, att2_weights, ground_weights, target, roi_labels[:, :T, :].clone(), input_seq[:, 1:T + 1, 0].clone())
        if self.opts.train_decoder_only:                                              # reference :297-307
            return lm_loss.reshape(1), att2_loss.reshape(1), ground_loss.reshape(1), cls_loss.reshape(1)

        # ---- argmax cut, Loop B: localize (no recurrence -> all T in one attention call)  :313-338
        loc_emb = self._embed(output_seq, "emb_b")                                   # [B, T, E]
        loc_pool, loc_conv, _prob = self.localizer_core.forward_all_steps(loc_emb, conv_feats, p_conv_feats, pool_feats,
                                                                           p_pool_feats, region_mask)

        # ---- Loop C: reconstruct from the localized regions (sequential)             reference :348-362
        state = self._init_step_state(B)
        emb_all_c = self._embed(gt_caption[:, :T], "emb_c") if self.training else emb_all      # fresh dropout mask in training
        rec_outputs = []
        emb_steps_c, pool_steps, conv_steps = emb_all_c.unbind(1), loc_pool.unbind(1), loc_conv.unbind(1)
        # loop C knows even more of its inputs beforehand: the localized context of every step (loop B's output)
        g_att_c = F_.hoisted_gates(w_att, att_segs(emb_all_c), T, B) if hoist else None
        ctx_all = (loc_pool + loc_conv) if hoist else None                                  # [B, T, R], one add instead of T
        g_lang_c = F_.hoisted_gates(self.attended_roi_decoder_core.lang_lstm.weight_ih, [(ctx_all, 0)], T, B) if hoist else None
        ctx_steps = ctx_all.unbind(1) if g_lang_c is not None else None
        for t in range(T):
            output, state = self.attended_roi_decoder_core.step(
                emb_steps_c[t], fc_feats, pool_steps[t], conv_steps[t], state, drop_site="out_c.%d" % t,
                gate_pre_att=None if g_att_c is None else g_att_c[t], gate_pre_lang=None if g_lang_c is None else g_lang_c[t],
                ctx_sum=None if ctx_steps is None else ctx_steps[t])
            rec_outputs.append(output)
        lm_recon_loss = self.xe_criterion.from_logits(self._logits(torch.stack(rec_outputs, 1).view(B * T, -1)), target)
        return (lm_loss.reshape(1), att2_loss.reshape(1), ground_loss.reshape(1), cls_loss.reshape(1),
                lm_recon_loss.reshape(1))

    # ------------------------------------------------------------------ training pass on the C-driven loops
    def _loop_plan(self, B, like):
        """(attention kind, 1 / temperature, dropout spec of loop A, of loop C) when the two recurrent loops of this pass can run
        as one C-driven autograd node each (cvc/train_loops.py), else None (per-step path: dictated dropout masks, library
        dropout, widths the packed kernels do not take).  More than 64 clips: groups of <= 64 (cvc.train_loops.clip_groups)."""
        from .. import train_loops
        from .modules import AdditiveSoftAttention
        dc, rc = self.decoder_core, self.attended_roi_decoder_core
        sa = dc.soft_attn
        R, E, A = self.rnn_size, self.embed[0].weight.shape[1], sa.h2attn.weight.shape[0]
        if not train_loops.eligible(B, R, E, A, like, self.seq_length) or dropout.active():
            return None
        if not (dc.att_lstm.weight_ih.is_contiguous() and dc.lang_lstm.weight_ih.is_contiguous() and sa.h2attn.weight.is_contiguous()):
            return None
        specs = []
        for mod, site in ((dc.dropout, "out_a.0"), (rc.dropout, "out_c.0")):
            if mod.training and mod.p > 0:
                if not (mod.p < 1 and dropout.in_kernel(like)):
                    return None
                specs.append((dropout.rng_state(like.device), dropout.site_id(site), float(mod.p)))
            else:
                specs.append(None)
        additive = isinstance(sa, AdditiveSoftAttention)
        return (hip.ATTN_ADDITIVE if additive else hip.ATTN_DOT, 1.0 if additive else 1.0 / float(sa.temp), specs[0], specs[1])

    def _forward_loops_driven(self, plan, emb_all, gt_caption, input_seq, fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats,
                              g_pool_feats, region_mask, step_fmask, frm_mask_output, roi_labels, cls_loss):
        """_forward_3_loops from loop A on (reference :242-382) with loops A and C as one autograd node each; rows of the loops'
        outputs are t-major (t * B + b), so the vocabulary head and its criterion run on t-major rows too."""
        from .. import train_loops
        attn_kind, inv_temp, drop_a, drop_c = plan
        B, T, R = emb_all.shape[0], self.seq_length, self.rnn_size
        num_rois = pool_feats.shape[1]
        dc = self.decoder_core
        fc_in = fc_feats if self.opts.global_img_in_attn_lstm else None
        # the loops take 64 clips: a larger batch runs them once per group of clips (clips are independent through both loops), each
        # group with its own arena; what is batch-wide below (head, criteria, grounder, localizer, embeddings) is one call over all B
        groups = train_loops.clip_groups(B)
        nslots = 1 if self.opts.train_decoder_only else 2
        arenas = [train_loops.LoopArena(nslots, T, b1 - b0, R, emb_all.shape[2], emb_all.device) for b0, b1 in groups]
        feats = (pool_feats, p_pool_feats, conv_feats, p_conv_feats)
        cut = (lambda x, b0, b1: x) if len(groups) == 1 else (lambda x, b0, b1: None if x is None else x[b0:b1])
        cut_t = (lambda x, b0, b1: x) if len(groups) == 1 else (lambda x, b0, b1: None if x is None else x[:, b0:b1].contiguous())
        join_t = (lambda xs: xs[0]) if len(groups) == 1 else (lambda xs: None if xs[0] is None else torch.cat(xs, 1))      # [T, b, .] -> [T, B, .]
        for a_ in arenas:
            a_.sole = len(groups) == 1           # several groups: a weight's gradient has several producers (see LoopArena.sole)

        def grp_drop(spec, g):
            """a group's own dropout sites (site0 + 64 g + t): without it clip i of every group would draw clip i's mask of group 0"""
            return None if spec is None else (spec[0], spec[1] + 64 * g, spec[2])

        parts = [train_loops.decode_loop(arenas[g], cut(emb_all, b0, b1), cut(fc_in, b0, b1), tuple(cut(f, b0, b1) for f in feats),
                                         cut(region_mask, b0, b1), cut_t(step_fmask, b0, b1), dc.att_lstm, dc.lang_lstm, dc.soft_attn,
                                         attn_kind, inv_temp, grp_drop(drop_a, g)) for g, (b0, b1) in enumerate(groups)]
        out_a = join_t([p_[0] for p_ in parts])
        fm = join_t([p_[1] for p_ in parts])
        target = gt_caption[:, 1:T + 1].clone()
        out_c, done = None, None
        head = self.critLM.head_words(out_a.view(T * B, R), self.logit, target, t_major=True) if all(a.joint_ok() for a in arenas) else None
        if head is not None:
            # A group's two loops fit the joint back-propagation (64 + 64 rows: config 3; 32 + 32: the shares of the 8-GPU job): loops B
            # and C come FIRST, fed the argmax words of the head's arithmetic, and loop A's outputs pass through loop C's graph node
            # -- its backward then holds both loops' output gradients and runs their back-propagation as one pass (cvc.train_loops._Loop).
            done, output_seq = head
            out_c, out_a, fm = self._loops_b_c(arenas, groups, output_seq, gt_caption, emb_all, fc_in, conv_feats, p_conv_feats, pool_feats,
                                               p_pool_feats, region_mask, attn_kind, drop_c, joint=[(p_[0], p_[1]) for p_ in parts])
        att2_weights = fm.transpose(0, 1)                                             # [B, T, N] pre-softmax (:273)

        # ---- grounder over all T                                                      reference :282-294
        xt_clamp = torch.clamp(input_seq[:, 1:T + 1, 0].clone() - self.vocab_size, min=0)
        xt_all = self._vis_embed(xt_clamp)
        if hasattr(self.roi_feat_extractor, 'vis_classifiers_bias'):
            bias = self.roi_feat_extractor.vis_classifiers_bias[xt_clamp].type(xt_all.type()).unsqueeze(2).expand(B, T, num_rois)
        else:
            bias = 0
        ground_weights = self._grounder(xt_all, g_pool_feats, frm_mask_output[:, :, 1:], bias + att2_weights)
        if self.debug_collect is not None:
            self.debug_collect.update(ground_weights=ground_weights, att2_weights=att2_weights, roi_labels=roi_labels,
                                      frm_mask_output=frm_mask_output)
        # vocabulary head + criterion + the argmax cut of :313 as one op on the t-major rows (no logits tensor)
        lm_loss, att2_loss, ground_loss, output_seq = self.critLM.from_head(
            out_a.view(T * B, R), self.logit, att2_weights, ground_weights, target, roi_labels[:, :T, :], input_seq[:, 1:T + 1, 0],
            t_major=True, done=done)
        if self.opts.train_decoder_only:                                              # reference :297-307
            return lm_loss.reshape(1), att2_loss.reshape(1), ground_loss.reshape(1), cls_loss.reshape(1)
        if out_c is None:
            out_c = self._loops_b_c(arenas, groups, output_seq, gt_caption, emb_all, fc_in, conv_feats, p_conv_feats, pool_feats, p_pool_feats,
                                    region_mask, attn_kind, drop_c)
        lm_recon_loss = self.xe_criterion.from_head(out_c.view(T * B, R), self.logit, target, t_major=True)
        return (lm_loss.reshape(1), att2_loss.reshape(1), ground_loss.reshape(1), cls_loss.reshape(1), lm_recon_loss.reshape(1))

    def _loops_b_c(self, arenas, groups, output_seq, gt_caption, emb_all, fc_in, conv_feats, p_conv_feats, pool_feats, p_pool_feats,
                   region_mask, attn_kind, drop_c, joint=None):
        """the argmax cut, loop B (localize, all T in one attention call; reference :313-338) and loop C (reconstruct from the
        localized regions; reference :348-362).  Loop B and the embeddings are one call over the whole batch; loop C runs once per
        group of <= 64 clips (cvc.train_loops.clip_groups).  -> out_c [T, B, R], or (out_c, loop A's out, loop A's fm) with `joint`
        (per group: loop A's (out, fm) of that group)"""
        from .. import train_loops
        dc, T = self.decoder_core, self.seq_length
        loc_emb = self._embed(output_seq, "emb_b")                                   # [B, T, E]
        ctx_all = self.localizer_core.forward_all_steps_sum(loc_emb, conv_feats, p_conv_feats, pool_feats, p_pool_feats, region_mask)
        emb_all_c = self._embed(gt_caption[:, :T], "emb_c") if self.training else emb_all      # fresh dropout mask in training
        one = len(groups) == 1
        outs = []
        for g, (b0, b1) in enumerate(groups):
            sl = (lambda x: x) if one else (lambda x: None if x is None else x[b0:b1])
            drop = None if drop_c is None else (drop_c[0], drop_c[1] + 64 * g, drop_c[2])
            outs.append(train_loops.recon_loop(arenas[g], sl(emb_all_c), sl(fc_in), sl(ctx_all), dc.att_lstm, dc.lang_lstm, dc.soft_attn,
                                               attn_kind, drop, joint=None if joint is None else joint[g]))
        if one:
            return outs[0]
        if joint is None:
            return torch.cat(outs, 1)
        fms = [o[2] for o in outs]
        return (torch.cat([o[0] for o in outs], 1), torch.cat([o[1] for o in outs], 1),
                None if fms[0] is None else torch.cat(fms, 1))

    def _vis_embed(self, xt_clamp):
        """roi_feat_extractor.vis_embed (Embedding -> ReLU -> Dropout, backbone.py:55-57) on the grounder's class indices"""
        ve = self.roi_feat_extractor.vis_embed
        ve_seq = self.training and isinstance(ve, nn.Sequential) and len(ve) == 3 and isinstance(ve[2], nn.Dropout)
        if ve_seq and 0 < ve[2].p < 1 and dropout.in_kernel(ve[0].weight) and isinstance(ve[0], nn.Embedding) and ve[0].weight.shape[1] % 4 == 0:
            # lookup + ReLU + dropout in the embedding kernel, mask generated there
            return F_.embed_relu(ve[0].weight, xt_clamp, rng=(dropout.rng_state(xt_clamp.device), dropout.site_id("vis_embed"), float(ve[2].p)))
        if dropout.active() and ve_seq:
            return dropout.apply(ve[2], ve[1](ve[0](xt_clamp)), "vis_embed")         # dictated mask (train-mode parity tests)
        return ve(xt_clamp)

    # ------------------------------------------------------------------ inference
    def decode_weights(self) -> DecodeWeights:
        """Flat (and, on first use by the packed decode path, fragment-packed) views of the hot-path parameters.
        Cached until any parameter is modified in place (optimizer step, load_state_dict), detected through the
        tensors' version counters, so an evaluation loop binds / packs the checkpoint once."""
        params = [p for _, p in sorted(self.state_dict(keep_vars=True).items())]
        key = tuple((p.data_ptr(), p._version) for p in params)
        cached = getattr(self, "_decode_cache", None)
        if cached is None or cached[0] != key:
            sd = {k: v for k, v in self.state_dict().items()}
            cached = (key, DecodeWeights(sd, getattr(self.opts, "softattn_type", "additive")))
            self._decode_cache = cached
        return cached[1]

    def invalidate_decode_cache(self):
        """Drop the cached decode binding.  Needed after parameter updates that bypass the tensors' version counters:
        a HIP-graph replay of the training step, or a fused optimizer kernel (Trainer calls this after every step)."""
        self._decode_cache = None
        self._engine_cache = None
        hip.bump_weights_generation()            # the encoder's packed GRU / dense operands are keyed on it (cvc/gru.py, cvc/dense.py)

    @torch.no_grad()
    def _sample(self, segs_feat, seq, proposals, gt_caption, num, mask_boxes, gt_boxes, region_feats, frm_mask, sample_idx,
                pnt_mask, beam_size: Optional[int] = None):
        """reference :384-443: exactly seq_length decoder steps from BOS, no EOS early exit, UNK
        suppressed; returns (seq [B,T], att2_weights [B,T,N] post-softmax, None)."""
        _ov, (fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, _g, pnt_mask, _o, _c, _l) = self._encode(
            segs_feat, proposals, num, mask_boxes, region_feats, gt_boxes, frm_mask, sample_idx, pnt_mask)
        feats = dict(fc_feats=fc_feats.contiguous(), conv_feats=conv_feats.contiguous(), p_conv_feats=p_conv_feats.contiguous(),
                     pool_feats=pool_feats.contiguous(), p_pool_feats=p_pool_feats.contiguous(), pnt_mask=pnt_mask)
        beam = self.beam_size if beam_size is None else int(beam_size)
        temp = float(getattr(self.opts, "softmax_temp", 1.0))
        weights = self.decode_weights()
        # one engine (bound launch list + captured graph) per batch shape, reused across the batches of an evaluation
        # loop: the next batch is copied into the engine's own feature buffers instead of re-binding and re-capturing
        key = (id(weights), tuple(fc_feats.shape), tuple(conv_feats.shape), tuple(pool_feats.shape), beam, temp, self.seq_length,
               self.use_hip_graph)
        cached = getattr(self, "_engine_cache", None)
        if cached is not None and cached[0] == key:
            engine = cached[1]
            # next batch of the same shape.  Driver-bound engine without a graph: the C-ABI plan is simply pointed at the new
            # tensors.  Graph replay (or the ring fallback for odd widths): the batch is copied into the engine's own buffers
            # (0.2 ms at cfg2) -- a captured graph keeps the pointers it was captured with.
            if engine._plan is not None and engine.graph is None:
                engine.bind_features(feats)
            else:
                engine.load_features(feats)
        else:
            engine = DecodeEngine(weights, feats, self.seq_length, self.unk_idx, beam=beam, inv_temp=1.0 / temp, own_features=True)
            if self.use_hip_graph:
                engine.capture()
            self._engine_cache = (key, engine)
        res = engine.run()
        return res[0].clone(), res[1].clone(), None
