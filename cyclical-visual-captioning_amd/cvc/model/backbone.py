"""Once-per-clip region / frame encoder -- SURVEY.md section 8(f) rank 1, the caller on the input
side of the hot path.  Mirrors the reference's `RegionalFeatureExtractorGVD`
(model/backbone.py:12-351): same constructor inputs (`opts` + the four Detectron pickles), same
`state_dict` keys, same 10-tuple returned to the captioner (backbone.py:350-351).

It runs once per clip (not once per decode step).  Its heavy pieces run on the build's own kernels: the 2-layer bidirectional
frame-context GRU (cvc/gru.py: tile-GEMM input projections + persistent recurrence, inference and autograd) and the dense layers
over the B*F frame rows / B*N region rows (cvc/dense.py on the tile GEMM); whatever falls outside their range (CPU tensors, an
nn.LSTM frame encoder, odd widths, a HIP graph under capture) uses the library module and says so once (cvc.hip.warn_once).
What differs from the reference's formulation:

  * the class-similarity logits are one `[DET+1, G] x [B, G, N]` product instead of a product
    against a per-clip expanded copy of the class table (backbone.py:222-233);
  * the pointer / sample-index masks are built with one broadcast compare instead of a Python loop
    with a host read per clip (backbone.py:191-203);
  * the region-classification loss is a masked mean, so the "no positive target" case needs no
    host-side branch (backbone.py:244-251); `cls_pred` (a data-dependent-length statistic) is only
    materialised when `opts.collect_cls_pred` is set (the reference computes it and never reads it:
    trainer.py:93-95 unpack only the losses).
"""
from __future__ import annotations

import os
import pickle

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import dense, encoder_ops, hip, gru as gru_hip

HIP_GRU = True     # inference: frame-context GRU on the HIP kernels (cvc/gru.py); False = the library module everywhere

GLOVE_DIM = 300          # backbone.py:43,46
SEG_INFO_SIZE = 50       # backbone.py:29
RGB_DIM, MOTION_DIM = 2048, 1024   # backbone.py:68,73,328
MIN_VALUE = -1e8         # backbone.py:40


def _load_pickle(directory, name, tables=None):
    if tables is not None:                      # in-memory tables (synthetic runs): same four arrays as the pickles
        return torch.as_tensor(tables[name[:-len(".pkl")]]).float()
    path = os.path.join(directory, name)
    if not os.path.exists(path):
        raise FileNotFoundError(
            "%s not found: RegionalFeatureExtractorGVD initialises its fc7 / class-score weights from the Detectron "
            "pickles (reference backbone.py:110-125); set opts.detectron_weights_dir" % path)
    with open(path, "rb") as f:
        return torch.from_numpy(pickle.load(f)).float()


def _relu_drop(layer, p):
    # the reference marks three of these Dropouts inplace (backbone.py:62-76); on this torch the ReLU in
    # front saves its OUTPUT for backward, so an in-place dropout trips autograd's version check -- same
    # values either way, so all are out-of-place here
    return nn.Sequential(layer, nn.ReLU(), nn.Dropout(p))


def project_and_mask(feat, projector, keep, site=None):
    """reference model/modules.py:162-176 (`proj_masking`) without its host-side assert."""
    out = dense.apply(projector, feat, site)
    return out * keep.unsqueeze(-1).to(out.dtype)


def match_visual_classifiers(glove_clss, glove_vg_cls, cls_score_w, cls_score_b):
    """Nearest Visual-Genome class (cosine over GloVe) for every detection class; row 0 (background)
    keeps VG row 0.  reference backbone.py:127-146."""
    a = glove_vg_cls / glove_vg_cls.norm(dim=1, keepdim=True)
    b = glove_clss / glove_clss.norm(dim=1, keepdim=True)
    matched = (a @ b.t()).argmax(dim=0)
    matched[0] = 0
    return cls_score_w[matched].clone(), cls_score_b[matched].clone()


class RegionalFeatureExtractorGVD(nn.Module):
    def __init__(self, opts):
        super().__init__()
        self.opts = opts
        self.test_mode = opts.test_mode
        self.enable_BUTD = opts.enable_BUTD
        self.att_input_mode = opts.att_input_mode
        self.num_sampled_frm = opts.num_sampled_frm
        self.rnn_size = R = opts.rnn_size
        self.seq_per_img = opts.seq_per_img
        self.seg_info_size = SEG_INFO_SIZE
        self.att_feat_size = opts.att_feat_size
        self.fc_feat_size = opts.fc_feat_size + SEG_INFO_SIZE
        self.detect_size = opts.detect_size
        self.pool_feat_size = opts.att_feat_size + GLOVE_DIM + opts.detect_size + 1
        self.vis_encoding_size = G = opts.vis_encoding_size
        self.t_attn_size = opts.t_attn_size
        self.collect_cls_pred = bool(getattr(opts, "collect_cls_pred", False))
        p = opts.drop_prob_lm

        self.loc_fc = _relu_drop(nn.Linear(5, GLOVE_DIM), p)
        self.det_fc = _relu_drop(nn.Embedding(opts.detect_size + 1, GLOVE_DIM), p)   # frozen GloVe table (:46-50)
        self.det_fc[0].weight.data.copy_(opts.glove_clss)
        self.det_fc[0].weight.requires_grad = False
        self.vis_embed = _relu_drop(nn.Embedding(opts.detect_size + 1, G), p)
        self.fc_embed = _relu_drop(nn.Linear(self.fc_feat_size, R), p)
        self.seg_info_embed = _relu_drop(nn.Linear(4, SEG_INFO_SIZE), p)
        self.att_embed = nn.ModuleList([_relu_drop(nn.Linear(RGB_DIM, R // 2), p),
                                        _relu_drop(nn.Linear(MOTION_DIM, R // 2), p)])
        self.att_embed_aux = nn.Sequential(nn.BatchNorm1d(R), nn.ReLU())
        self.pool_embed = _relu_drop(nn.Linear(self.pool_feat_size, R), opts.second_drop_prob)
        self.ctx2att_fc = nn.Linear(R, opts.att_hid_size)
        self.ctx2pool_fc = nn.Linear(R, opts.att_hid_size)
        if opts.att_model == 'transformer':
            raise NotImplementedError()
        rnn = {'bilstm': nn.LSTM, 'bigru': nn.GRU}.get(opts.t_attn_mode)
        if rnn is None:
            raise NotImplementedError
        self.context_enc = rnn(R, R // 2, 2, dropout=0.2, bidirectional=True, batch_first=True)
        self.ctx2pool_grd = _relu_drop(nn.Linear(opts.att_feat_size, G), p)           # Detectron fc7

        wdir = getattr(opts, "detectron_weights_dir", os.path.join("data", "detectron_weights"))
        tables = getattr(opts, "detectron_tables", None)
        n = opts.att_feat_size
        with torch.no_grad():
            self.ctx2pool_grd[0].weight[:n].copy_(_load_pickle(wdir, "fc7_w.pkl", tables))
            self.ctx2pool_grd[0].bias[:n].copy_(_load_pickle(wdir, "fc7_b.pkl", tables))
        cls_w, cls_b = _load_pickle(wdir, "cls_score_w.pkl", tables), _load_pickle(wdir, "cls_score_b.pkl", tables)
        assert len(opts.itod) + 1 == opts.glove_clss.size(0)
        assert len(opts.vg_cls) == opts.glove_vg_cls.size(0)
        w, b = match_visual_classifiers(opts.glove_clss, opts.glove_vg_cls, cls_w, cls_b)
        self.vis_classifiers_bias = nn.Parameter(b)
        self.vis_embed[0].weight.data.copy_(w)

    # ------------------------------------------------------------------ once-per-clip features
    def class_similarity(self, g_pool_feats, pad):
        """softmax over detection classes of <class classifier, region feature> + class bias, padded
        regions filled with -1e8 before the softmax.  backbone.py:216-235."""
        from .. import dropout
        ve = self.vis_embed
        table = dropout.apply(ve[2], ve[1](ve[0](torch.arange(self.detect_size + 1, device=g_pool_feats.device))), "enc.vis_table")
        sim = torch.matmul(table, g_pool_feats.transpose(1, 2)) + self.vis_classifiers_bias.view(1, -1, 1)
        return F.softmax(sim.masked_fill(pad.unsqueeze(1), MIN_VALUE), dim=1)

    def region_class_loss(self, sim, overlaps, gt_boxes):
        """BCE against 1 of the similarity at (gt class, region) wherever IoU > 0.5.  backbone.py:240-262."""
        target = ((overlaps > 0.5).long() * gt_boxes[:, :, 5].long().unsqueeze(1)).transpose(1, 2).contiguous()
        pos = target > 0
        picked = torch.gather(sim, 1, target)
        nll = -(torch.log(picked).clamp(min=-100.0))
        cnt = pos.sum()
        loss = (nll * pos).sum() / cnt.clamp(min=1).to(nll.dtype)
        pred = None
        if self.collect_cls_pred:
            arg = sim.argmax(dim=1).unsqueeze(1).expand_as(target)
            pred = torch.stack((target[pos], arg[pos]), dim=1)
        return loss, pred

    def get_conv_pooled_feats(self, segs_feat, proposals, mask_boxes, num, region_feats, gt_boxes, overlaps, sample_idx,
                              eval_obj_ground=False, replicate_feat=True):
        """reference backbone.py:178-294"""
        B, N = segs_feat.size(0), proposals.size(1)
        S = self.seq_per_img
        dev = segs_feat.device
        col = torch.arange(N + 1, device=dev).unsqueeze(0)
        pnt_mask = col > num[:, 1].long().to(dev).unsqueeze(1)                        # slot 0 = sentinel ROI
        frame = torch.arange(segs_feat.size(1), device=dev).unsqueeze(0)
        si = sample_idx.to(dev).long()
        sample_idx_mask = ((frame < si[:, :1]) | (frame >= si[:, 1:2])).unsqueeze(-1)
        keep = ~pnt_mask[:, 1:]

        fc = segs_feat.mean(dim=1)
        seg_info = dense.apply(self.seg_info_embed, num[:, 3:7].float(), "enc.seg_info")
        # inference on the GPU: the fused kernels of csrc/encoder_ops.hip; under autograd: their training forms (csrc/encoder_train.hip)
        fused = encoder_ops.usable(region_feats, self) and segs_feat.dtype == torch.float32
        fused_t = (not fused) and encoder_ops.usable_train(region_feats) and segs_feat.dtype == torch.float32
        if not (fused or fused_t) and region_feats.is_cuda:
            hip.warn_once("encoder-ops-library", "the encoder's class-similarity softmax / layer norms / frame-embedding epilogue run "
                          "on library kernels (non-fp32 input, dictated dropout masks, or train() mode without autograd)")
        if fused:
            fc_feats = encoder_ops.layernorm_cat([fc.float(), seg_info])
        elif fused_t:
            fc_feats = encoder_ops.layernorm_cat_train([fc.float(), seg_info])
        else:
            fc_feats = torch.cat((F.layer_norm(fc, [self.fc_feat_size - SEG_INFO_SIZE]),
                                  F.layer_norm(seg_info, [SEG_INFO_SIZE])), dim=-1)

        g_pool_feats = project_and_mask(region_feats, self.ctx2pool_grd, keep, "enc.ctx2pool_grd")
        sim_rows = None
        if fused:
            sim, sim_rows = encoder_ops.class_similarity(self, g_pool_feats, pnt_mask[:, 1:])
        elif fused_t:
            sim, sim_rows = encoder_ops.class_similarity_train(self, g_pool_feats, pnt_mask[:, 1:])
        else:
            sim = self.class_similarity(g_pool_feats, pnt_mask[:, 1:])

        if self.test_mode:
            cls_pred, cls_loss = 0, torch.zeros(1, device=dev)
        else:
            cls_loss, cls_pred = self.region_class_loss(sim, overlaps, gt_boxes)

        pool_feats = g_pool_feats
        if not self.enable_BUTD:
            loc = torch.cat((proposals[:, :, :4] / 720., proposals[:, :, 4:5] / float(self.num_sampled_frm)), dim=2)
            loc_feats = dense.apply(self.loc_fc, loc.detach(), "enc.loc_fc")
            if fused:
                pool_feats = encoder_ops.layernorm_cat([g_pool_feats, loc_feats, sim_rows])
            elif fused_t:
                pool_feats = encoder_ops.layernorm_cat_train([g_pool_feats, loc_feats, sim_rows])
            else:
                label_feat = sim.transpose(1, 2)
                pool_feats = torch.cat((F.layer_norm(g_pool_feats, [g_pool_feats.size(-1)]),
                                        F.layer_norm(loc_feats, [GLOVE_DIM]),
                                        F.layer_norm(label_feat, [label_feat.size(-1)])), dim=2)

        def per_caption(x):
            return x if S == 1 else x.repeat_interleave(S, dim=0)

        return (per_caption(fc_feats), segs_feat, per_caption(pool_feats), per_caption(g_pool_feats),
                per_caption(pnt_mask), per_caption(overlaps), sample_idx_mask, cls_pred, cls_loss)

    def reports_error_words(self) -> bool:
        """the frame-context GRU's persistent forms (csrc/gru_persistent.hip, gru_bwd_persistent.hip) raise an error word on a
        barrier time-out"""
        return HIP_GRU and isinstance(self.context_enc, nn.GRU) and self.att_input_mode in ('both', 'featmap')

    def step_capturable(self, deferred_errors: bool = False) -> bool:
        """Can a training step through this encoder be captured into a HIP graph?  Everything here is stream work -- dense layers on
        the tile GEMM, the fused encoder kernels, the GRU's recurrences -- except: (i) the persistent recurrences' error words, read
        by the host unless the caller runs in deferred mode; (ii) the library fallbacks (an nn.LSTM frame encoder or a GRU width
        outside H % 8 == 0 runs in MIOpen, whose workspace handling is not ours to capture); (iii) collect_cls_pred's boolean
        gathers (host-sized outputs)."""
        if self.collect_cls_pred:
            return False
        if self.att_input_mode not in ('both', 'featmap'):
            return True                                        # no frame path, no recurrence
        enc = self.context_enc
        if not (HIP_GRU and isinstance(enc, nn.GRU) and enc.batch_first and enc.bias and enc.hidden_size % 8 == 0):
            return False
        return bool(deferred_errors)

    def _frame_context(self, x):
        """The 2-layer bidirectional frame-context RNN (backbone.py:103-106, 335-338): HIP kernels where they apply, else the
        library module -- and then says why, once."""
        enc = self.context_enc
        # (a capture may contain the HIP recurrence only in deferred mode: no host read of the persistent forms' error words)
        capturing = x.is_cuda and torch.cuda.is_current_stream_capturing() and not hip.errors_deferred()
        if HIP_GRU and not torch.is_grad_enabled() and gru_hip.supported(enc, x):
            return gru_hip.gru_forward(enc, x)                        # inference: persistent / per-step recurrence + tile GEMM
        if HIP_GRU and torch.is_grad_enabled() and gru_hip.supported_train(enc, x) and not capturing:
            try:
                return gru_hip.gru_forward_train(enc, x)              # autograd on the same kernels (cvc_gru_seq_bwd)
            except gru_hip.GruUnavailable as e:                       # barrier time-out / grid not co-resident: keep training
                hip.warn_once("gru-train-unavailable", f"frame-context GRU falls back to the library module for this run: {e}")
        elif x.is_cuda and HIP_GRU:
            why = ("a HIP graph is being captured (the persistent recurrence reports time-outs through a host read)" if capturing else
                   f"{type(enc).__name__} / hidden size {enc.hidden_size} is outside the HIP recurrence's range "
                   "(nn.GRU, batch_first, bias, H % 8 == 0)")
            hip.warn_once("gru-library:" + why[:24], "frame-context RNN runs on the library module (MIOpen): " + why)
        enc.flatten_parameters()
        return enc(x)[0]

    def forward(self, segs_feat, proposals, num, mask_boxes, region_feats, gt_boxes, overlaps, sample_idx,
                eval_obj_ground=False, replicate_feat=True):
        """reference backbone.py:296-351"""
        (fc_feats, conv_feats, pool_feats, g_pool_feats, pnt_mask, overlaps_expanded, sample_idx_mask, cls_pred,
         cls_loss) = self.get_conv_pooled_feats(segs_feat, proposals, mask_boxes, num, region_feats, gt_boxes, overlaps,
                                                sample_idx, eval_obj_ground, replicate_feat)
        keep = ~pnt_mask[:, 1:]
        fc_feats = dense.apply(self.fc_embed, fc_feats, "enc.fc_embed")
        pool_feats = project_and_mask(pool_feats, self.pool_embed, keep, "enc.pool_embed")
        p_pool_feats = project_and_mask(pool_feats, self.ctx2pool_fc, keep)

        if self.att_input_mode in ('both', 'featmap'):
            rgb, motion = conv_feats[..., :RGB_DIM], conv_feats[..., RGB_DIM:RGB_DIM + MOTION_DIM]
            if encoder_ops.usable(conv_feats, self) and dense.usable(rgb) and (self.rnn_size // 2) % 4 == 0:
                x = encoder_ops.frame_embed(self, rgb, motion)              # GEMMs + one fused epilogue (BN folded)
            elif (encoder_ops.usable_train(conv_feats) and self.att_embed_aux[0].training and (self.rnn_size // 2) % 4 == 0
                  and rgb.shape[-1] % 4 == 0 and motion.shape[-1] % 4 == 0):
                x = encoder_ops.frame_embed_train(self, rgb, motion)        # train(): in-kernel dropout, BatchNorm on batch statistics
            else:
                x = torch.cat((dense.apply(self.att_embed[0], rgb, "enc.att0"), dense.apply(self.att_embed[1], motion, "enc.att1")), dim=2)
                x = self.att_embed_aux(x.transpose(1, 2)).transpose(1, 2).contiguous()      # BatchNorm1d over channels
            x = self._frame_context(x)
            x = x.masked_fill(sample_idx_mask, 0)
            conv_feats = x if self.seq_per_img == 1 else x.repeat_interleave(self.seq_per_img, dim=0)
            p_conv_feats = dense.apply(self.ctx2att_fc, conv_feats)
        else:
            conv_feats = pool_feats.new_zeros(1, 1)
            p_conv_feats = pool_feats.new_zeros(1, 1)
        return (fc_feats, conv_feats, p_conv_feats, pool_feats, p_pool_feats, g_pool_feats, pnt_mask, overlaps_expanded,
                cls_pred, cls_loss)
