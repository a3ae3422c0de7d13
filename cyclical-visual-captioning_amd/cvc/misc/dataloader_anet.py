"""On-disk data path of the ActivityNet-Entities loader -- what the reference's `misc/dataloader_anet.py` (class DataLoader,
constructor :28-153, item assembly :185-388) reads and returns, rebuilt as small numpy steps:

  files  region features  `<feature_root>/<segment id>.npy`       [num_sampled_frm, num_prop_per_frm, att_feat_size] fp32
         frame features   `<seg_feature_root>/<video>_resnet.npy`  [n, 2048] and `<video>_bn.npy` [n, 1024]
         proposals        `dets_num` [segments], `dets_labels` [segments, max_proposal, 7] = (x1, y1, x2, y2, frame, class,
                          score) from the reference's `.h5` (needs h5py) or an `.npz` twin holding the same two arrays
         annotations      input_dic (vocabulary, class / lemma maps, segment list + splits), input_json (captions with
                          grounded boxes), grd_reference (segment timestamps, video durations)
  item   the 12-tuple of SURVEY.md section 3.4: frame features [t_attn_size, 3072] f64, input_seq [S, T+1, 4] i64,
         gt_seq [10, T] i64, num [7] f32, proposals [P, 7] f32, gt boxes [100, 6] f32, box mask [S, 100, T+1] bool,
         segment id, region features [P, att_feat_size] f32, frame mask [P, 100] bool, sampled-frame window [2] i64,
         proposal mask [P] bool, with P = num_sampled_frm * num_prop_per_frm.

Trainer._prepare trims a batch to its largest proposal / box count (reference trainer.py:63-69) and
cvc.prefetch.DevicePrefetcher stages it into HBM one batch ahead.

Not carried over: torchtext / torchvision / PIL imports.  GloVe comes from `opt.glove` (anything with `.stoi` and `.vectors`,
e.g. load_glove('glove.6B.300d.npz')); the Visual Genome class list from `opt.vg_vocab_file` (default: the reference's
hard-coded 'data/vg_object_vocab.txt').  The values of every tuple member and of the GloVe tables are pinned bit for bit
against the reference loader itself, run over a tiny synthetic dataset written in the reference's formats
(tools/make_golden.py g6 -> tests/golden/g6_dataloader.npz; tests/test_dataloader.py).
"""
from __future__ import annotations

import json
import os
import random
from collections import defaultdict

import numpy as np
import torch
import torch.utils.data as data

MAX_GT_BOX = 100            # reference :47


class GloveTable:
    """What the loader needs of torchtext.vocab.GloVe: `.stoi` (word -> row) and `.vectors` ([n, 300] tensor)."""

    def __init__(self, words, vectors):
        self.stoi = {w: i for i, w in enumerate(words)}
        self.vectors = torch.as_tensor(np.asarray(vectors), dtype=torch.float32)


def load_glove(path: str) -> GloveTable:
    """`.npz` with `words` [n] and `vectors` [n, 300]."""
    z = np.load(path, allow_pickle=False)
    return GloveTable([str(w) for w in z["words"]], z["vectors"])


def read_proposals(path: str):
    """(dets_num, dets_labels) of the proposal file: `.h5` as the reference stores it (:99-104) or an `.npz` twin."""
    if path.endswith(".npz"):
        with np.load(path) as z:
            return z["dets_num"][:], z["dets_labels"][:]
    try:
        import h5py
    except ImportError as e:
        raise ImportError(f"{path}: the proposal .h5 needs h5py; convert it once to an .npz holding 'dets_num' and "
                          "'dets_labels' and pass that as opt.proposal_h5") from e
    with h5py.File(path, "r", driver="core") as f:
        return f["dets_num"][:], f["dets_labels"][:]


def _mean_glove(glove, phrase_words):
    """Average GloVe vector of a phrase; unknown words draw a uniform(-1, 1) vector from numpy's global RNG, in phrase order
    (reference :76-87 / :123-132 -- the draw order is part of what the fixture pins)."""
    acc = np.zeros(300)
    for w in phrase_words:
        acc += glove.vectors[glove.stoi[w]].numpy() if w in glove.stoi else 2 * np.random.rand(300) - 1
    return acc / len(phrase_words)


class ANetEntitiesDataset(data.Dataset):
    def __init__(self, opt, split='training', seq_per_img=5, num_proposals=None, label_proposals=None):
        self.opt, self.split, self.seq_per_img = opt, split, seq_per_img
        for k in ("batch_size", "seq_length", "att_feat_size", "feature_root", "seg_feature_root", "num_sampled_frm",
                  "num_prop_per_frm", "exclude_bgd_det", "prop_thresh", "t_attn_size", "test_mode"):
            setattr(self, k, getattr(opt, k))
        self.max_gt_box = MAX_GT_BOX
        self.max_proposal = self.num_sampled_frm * self.num_prop_per_frm
        glove = getattr(opt, "glove", None)
        if glove is None:
            if not getattr(opt, "glove_path", ""):
                raise ValueError("ANetEntitiesDataset needs GloVe vectors: opt.glove (object with .stoi / .vectors) or "
                                 "opt.glove_path (.npz with 'words', 'vectors'); torchtext is not a dependency")
            glove = load_glove(opt.glove_path)
        self.glove = glove

        # ---- vocabulary and class maps (reference :51-61).  ix_to_word keys are strings; wtoi therefore maps to strings
        info = self.info = json.load(open(opt.input_dic))
        self.itow = info['ix_to_word']
        self.wtoi = {w: i for i, w in self.itow.items()}
        self.wtod = {w: i + 1 for w, i in info['wtod'].items()}
        self.dtoi = self.wtod
        self.itod = {i: w for w, i in self.wtod.items()}
        self.itoc = self.itod
        self.wtol = info['wtol']
        self.ltow = {l: w for w, l in self.wtol.items()}
        self.vocab_size = len(self.itow) + 1
        self.detect_size = len(self.itod)

        # ---- GloVe tables in the reference's order of RNG draws: VG classes, background class, detection classes, words
        with open(getattr(opt, "vg_vocab_file", 'data/vg_object_vocab.txt')) as f:
            self.vg_cls = ['__background__'] + [line.strip() for line in f]
        self.glove_vg_cls = np.stack([_mean_glove(glove, c.replace(',', ' ').split(' ')) for c in self.vg_cls])
        self.caption_file = json.load(open(opt.input_json))
        self.timestamp_file = self.grd_reference = json.load(open(opt.grd_reference))          # (also read by Trainer.eval)
        if num_proposals is None or label_proposals is None:
            num_proposals, label_proposals = read_proposals(opt.proposal_h5)
        self.num_proposals, self.label_proposals = num_proposals, label_proposals
        self.glove_clss = np.zeros((self.detect_size + 1, 300))
        self.glove_clss[0] = 2 * np.random.rand(300) - 1                                        # background (:111)
        for r, word in enumerate(self.itod.values(), start=1):
            self.glove_clss[r] = glove.vectors[glove.stoi[word]].numpy() if word in glove.stoi else 2 * np.random.rand(300) - 1
        self.glove_w = np.zeros((len(self.wtoi) + 1, 300))
        for r, word in enumerate(self.wtoi.keys(), start=1):
            self.glove_w[r] = _mean_glove(glove, word.split(' '))

        # ---- segments of this split whose feature files exist (:137-153)
        self.split_ix, self.num_seg_per_vid = [], defaultdict(list)
        for ix, seg in enumerate(info['videos']):
            vid, seg_no = seg['id'].split('_segment_')
            self.num_seg_per_vid[vid].append(int(seg_no))
            if seg['split'] == split and os.path.isfile(os.path.join(self.feature_root, seg['id'] + '.npy')) and \
                    os.path.isfile(os.path.join(self.seg_feature_root, vid[2:] + '_bn.npy')):
                self.split_ix.append(ix)

    def __len__(self):
        return len(self.split_ix)

    # ------------------------------------------------------------------ pieces of one item
    def _frames(self, vid, t0, t1, dur):
        """frame features [t_attn_size, 3072] (rgb | motion, zero padded / cut) and the sampled-frame window (:212-229;
        the window's arithmetic is kept operation for operation: n * t * 1. / dur, rounded twice)"""
        raw = np.concatenate([np.load(os.path.join(self.seg_feature_root, vid[2:] + s)) for s in ('_resnet.npy', '_bn.npy')], axis=1)
        n = raw.shape[0]
        idx = np.array([np.round(n * t0 * 1. / dur), np.round(n * t1 * 1. / dur)])
        idx = np.clip(np.round(idx), 0, self.t_attn_size).astype(int)
        out = np.zeros((self.t_attn_size, raw.shape[1]))
        out[:min(self.t_attn_size, n)] = raw[:self.t_attn_size]
        return out, idx

    def _grounded_boxes(self, cap):
        """rows (x1, y1, x2, y2, frame, class id, box id, word index) of the caption's boxes whose word lies inside
        seq_length, ordered by word index (stable), zero-area boxes dropped outside test mode (:236-270); plus the per-word
        (class + vocab offset, binary, fine) indicator of reference get_det_word (:155-174)."""
        rows, names = [], []
        for b, labels in enumerate(cap['clss']):
            for j, cls in enumerate(labels):
                w = cap['idx'][b][j]
                if w < self.seq_length:
                    box, frm = ([0, 0, 0, 0], -1) if self.test_mode else (cap['bbox'][b], cap['frm_idx'][b])
                    rows.append(list(box) + [frm, self.dtoi[cls], len(rows), w])
                    names.append(cls)
        order = sorted(range(len(rows)), key=lambda r: rows[r][7])
        boxes = np.asarray([rows[r] for r in order], dtype=np.float64).reshape(-1, 8)
        names = [names[r] for r in order]
        keep = np.ones(len(boxes), dtype=bool)
        if not self.test_mode:
            keep = (boxes[:, 2] - boxes[:, 0] + 1 != 1) & (boxes[:, 3] - boxes[:, 1] + 1 != 1)
        words = cap['caption']
        indicator = np.zeros((len(words), 3))
        for r in np.nonzero(keep)[0]:                 # later boxes on the same word overwrite earlier ones, as in the reference
            w = int(boxes[r, 7])
            indicator[w] = (self.wtod[names[r]], (names[r] != words[w]) + 1, boxes[r, 5])
        return boxes[keep], indicator

    def _caption_rows(self, words, indicator):
        """[seq_length, 5]: (token or class + vocab_size, binary, fine, token if grounded, token) per word (:280-295)"""
        n = min(len(words), self.seq_length)
        tok = np.asarray([int(self.wtoi[w]) for w in words[:n]], dtype=np.float64)
        rows = np.zeros((self.seq_length, 5))
        det = indicator[:n, 0] != 0
        rows[:n, 0] = np.where(det, indicator[:n, 0] + self.vocab_size, tok)
        rows[:n, 1] = np.where(det, indicator[:n, 1], 0)
        rows[:n, 2] = np.where(det, indicator[:n, 2], 0)
        rows[:n, 3] = np.where(det, tok, 0)
        rows[:n, 4] = tok
        return rows

    def __getitem__(self, index):
        ix = self.split_ix[index]
        seg_id = self.info['videos'][ix]['id']
        vid, seg_no = seg_id.split('_segment_')
        seg_no = str(int(seg_no))

        # proposals + region features, low-confidence / background proposals masked (:195-208)
        n_prop = int(self.num_proposals[ix])
        props = np.array(self.label_proposals[ix][:n_prop], copy=True)          # file dtype kept: the threshold compare runs in it
        region = np.load(os.path.join(self.feature_root, seg_id + '.npy'))
        region = region.reshape(-1, region.shape[2]).copy()
        assert n_prop == region.shape[0], (seg_id, n_prop, region.shape)
        drop = props[:, 6] <= self.prop_thresh
        if self.exclude_bgd_det:
            drop |= props[:, 5] == 0

        ann = self.timestamp_file['annotations'][vid]
        t0, t1 = ann['segments'][seg_no]['timestamps']
        dur = ann['duration']
        frames, sample_idx = self._frames(vid, t0, t1, dur)

        cap = self.caption_file[vid]['segments'][seg_no]                                        # one caption per segment (:232-234)
        boxes, indicator = self._grounded_boxes(cap)
        rows = self._caption_rows(cap['caption'], indicator)[None]                              # [ncap = 1, T, 5]
        ncap = 1
        word_mask = np.ones((ncap, len(boxes), self.seq_length))                                # False where box b grounds word t
        word_mask[0, np.arange(len(boxes)), boxes[:, 7].astype(int)] = 0

        # seq_per_img captions out of ncap (:306-318; the RNG call pattern is the reference's)
        if ncap < self.seq_per_img:
            pick = [random.randint(0, ncap) for _ in range(self.seq_per_img)]
        else:
            first = random.randint(0, ncap - self.seq_per_img)
            pick = list(range(first, first + self.seq_per_img))
        input_seq = np.zeros((self.seq_per_img, self.seq_length + 1, 4))
        input_seq[:, 1:] = rows[pick][:, :, :4]
        gt_seq = np.zeros((10, self.seq_length))
        gt_seq[:ncap] = rows[:, :, 4]

        # fixed-size padding (:343-363)
        P, K = self.max_proposal, self.max_gt_box
        n_pps, n_box = min(n_prop, P), min(len(boxes), K)
        pad_props = np.zeros((P, 7)); pad_props[:n_pps] = props[:n_pps]
        pad_drop = np.ones(P); pad_drop[:n_pps] = drop[:n_pps]
        pad_boxes = np.zeros((K, 6)); pad_boxes[:n_box] = boxes[:n_box, :6]
        pad_word_mask = np.ones((self.seq_per_img, K, self.seq_length + 1)); pad_word_mask[:, :n_box, 1:] = word_mask[pick][:, :n_box]
        pad_region = np.zeros((P, self.att_feat_size)); pad_region[:n_pps] = region[:n_pps]
        pad_frm = np.ones((P, K))
        pad_frm[:n_pps, :n_box] = pad_props[:n_pps, 4:5] != pad_boxes[None, :n_box, 4]          # proposal frame != box frame

        drop_t = torch.from_numpy(pad_drop).bool()
        props_t = torch.from_numpy(pad_props).float().masked_fill_(drop_t.view(-1, 1), 0.)
        region_t = torch.from_numpy(pad_region).float().masked_fill_(drop_t.view(-1, 1), 0.)
        num = torch.FloatTensor([ncap, n_pps, n_box, int(seg_no), max(self.num_seg_per_vid[vid]) + 1, t0 * 1. / dur, t1 * 1. / dur])
        return (frames, torch.from_numpy(input_seq).long(), torch.from_numpy(gt_seq).long(), num, props_t,
                torch.from_numpy(pad_boxes).float(), torch.from_numpy(pad_word_mask).bool(), seg_id, region_t,
                torch.from_numpy(pad_frm).bool(), torch.from_numpy(sample_idx).long(), drop_t)


DataLoader = ANetEntitiesDataset          # the reference's class name (main.py:78-96 constructs `DataLoader(opt, split=...)`)


def collate(items):
    """default_collate for the 12-tuple: numpy frame features -> one float64 tensor, ids -> list, tensors -> stack."""
    out = []
    for col in zip(*items):
        if isinstance(col[0], str):
            out.append(list(col))
        elif isinstance(col[0], np.ndarray):
            out.append(torch.from_numpy(np.stack(col)))
        else:
            out.append(torch.stack(col))
    return tuple(out)
