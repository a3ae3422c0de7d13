"""A tiny synthetic ActivityNet-Entities dataset written in the REFERENCE's on-disk formats (misc/dataloader_anet.py): the
annotation JSONs, per-segment region-feature `.npy`, per-video frame-feature `.npy` pairs, the proposal arrays (as `.npz`;
the reference reads the same two arrays from an `.h5`), a Visual Genome class list and a GloVe table.  Deterministic in
`seed`.  Used by tools/make_golden.py (reference loader -> fixture), tests/test_dataloader.py (this repo's loader vs the
fixture) and as a runnable example of the data layout; covers the loader's corner cases: multi-label boxes, a box past
seq_length, a zero-area box, ragged proposal counts, low-score / background proposals, clips longer and shorter than
t_attn_size, out-of-vocabulary GloVe words, and -- in the last video (`extreme`) -- a segment that overflows every padded
array of the loader (dataloader_anet.py:342-363): more proposals than num_sampled_frm * num_prop_per_frm, more than 100 ground-
truth boxes, a caption longer than seq_length."""
from __future__ import annotations

import json
import os
from types import SimpleNamespace

import numpy as np

from . import synth

WORDS = ["UNK", "a", "man", "woman", "dog", "ball", "rides", "throws", "the", "red", "bike", "field", "in", "and", "frisbee", "traffic light"]
CLASSES = ["man", "woman", "dog", "ball", "bike", "frisbee"]
VG_CLASSES = ["man", "dog,puppy", "ball", "zebra crossing", "bike"]


def write_tiny_anet_dataset(root: str, seed: int = 7, n_videos: int = 3, frames: int = 2, props_per_frame: int = 5,
                            feat: int = 12, seq_length: int = 8, t_attn: int = 6, rgb_dim: int = 5, bn_dim: int = 3,
                            extreme: bool = True) -> SimpleNamespace:
    """rgb_dim / bn_dim: widths of the two frame-feature files (2048 / 1024 in the real dataset; the mirrored encoder
    hard-codes that split as the reference does, backbone.py:68,73)."""
    os.makedirs(root, exist_ok=True)
    fr, sr = os.path.join(root, "region"), os.path.join(root, "seg")
    os.makedirs(fr, exist_ok=True); os.makedirs(sr, exist_ok=True); os.makedirs(os.path.join(root, "data"), exist_ok=True)
    itow = {str(i + 1): w for i, w in enumerate(WORDS)}
    wtod = {c: i for i, c in enumerate(CLASSES)}                       # reference adds 1 (:56)
    wtol = {w: w for w in WORDS}
    videos, caps, grd = [], {}, {"annotations": {}}
    P = frames * props_per_frame
    P_file = P + 3                                                        # rows of the proposal array (the real file has 1000)
    dets_num, dets_labels = [], []
    k = 0
    for v in range(n_videos + (1 if extreme else 0)):
        vid = "v_vid%02d" % v
        over = extreme and v == n_videos                                  # the overflowing segment
        n_seg = 1 if over else 1 + v % 2
        nfrm = (3, 9, 6)[v % 3]                                          # shorter / longer / equal to t_attn
        np.save(os.path.join(sr, vid[2:] + "_resnet.npy"), synth.normal((nfrm, rgb_dim), seed, "rgb%d" % v))
        np.save(os.path.join(sr, vid[2:] + "_bn.npy"), synth.normal((nfrm, bn_dim), seed, "bn%d" % v))
        caps[vid] = {"segments": {}}
        grd["annotations"][vid] = {"duration": 30.0 + 7.5 * v, "segments": {}}
        for s in range(n_seg):
            seg_id = "%s_segment_%02d" % (vid, s)
            split = "training" if (k % 4) != 3 else "validation"
            videos.append({"id": seg_id, "split": split})
            n_words = seq_length + 3 if over else 4 + (k * 3) % 6
            wi = synth.randint((n_words,), seed, "words%d" % k, 0, len(WORDS))
            words = [WORDS[i] for i in wi]
            words[1], words[3] = "man", "dog"                           # groundable words at fixed slots
            # boxes: multi-label, one past seq_length, one with zero area, one on word 1 / 3
            bbox = [[10, 20, 110, 220], [5, 5, 5, 5], [30, 40, 90, 140], [1, 2, 50, 60]]
            clss = [["man"], ["ball"], ["dog", "woman"], ["bike"]]
            idx = [[1], [2], [3, 3], [seq_length + 2]]
            frm = [0, 1, k % frames, 1]
            if over:                                                    # 113 boxes, every word slot the loader keeps in turn:
                for j in range(109):                                    # more than max_gt_box = 100 survive its filters
                    x0, y0 = 3 + 2 * j, 7 + j
                    bbox.append([x0, y0, x0 + 20 + j % 7, y0 + 30 + j % 5])
                    clss.append([CLASSES[j % len(CLASSES)]])
                    idx.append([j % seq_length])
                    frm.append(j % frames)
            caps[vid]["segments"][str(s)] = {"caption": words, "clss": clss, "idx": idx, "bbox": bbox, "frm_idx": frm}
            t0 = 2.0 + 3.3 * s
            grd["annotations"][vid]["segments"][str(s)] = {"timestamps": [t0, t0 + 6.1 + v]}
            n_p = P + 2 if over else P - (k % 3)                        # more proposals than num_sampled_frm * num_prop_per_frm
            lab = np.zeros((P_file, 7), dtype=np.float32)
            xy = synth.uniform((n_p, 2), seed, "xy%d" % k, 0, 200)
            wh = synth.uniform((n_p, 2), seed, "wh%d" % k, 10, 100)
            lab[:n_p, 0:2], lab[:n_p, 2:4] = xy, xy + wh
            lab[:n_p, 4] = np.minimum(np.arange(n_p) // props_per_frame, frames - 1)
            lab[:n_p, 5] = synth.randint((n_p,), seed, "cls%d" % k, 0, len(CLASSES) + 1)      # 0 = background
            lab[:n_p, 6] = synth.uniform((n_p,), seed, "sc%d" % k, 0.05, 1.0)
            lab[0, 6] = np.float32(0.2)                                  # exactly at the threshold
            dets_num.append(n_p); dets_labels.append(lab)
            rf = synth.normal((n_p, feat), seed, "region%d" % k)
            # the file holds exactly num_proposal rows once flattened (reference asserts it, :202)
            np.save(os.path.join(fr, seg_id + ".npy"), rf.reshape(1, n_p, feat))
            k += 1
    json.dump({"ix_to_word": itow, "wtod": wtod, "wtol": wtol, "videos": videos}, open(os.path.join(root, "dic.json"), "w"))
    json.dump(caps, open(os.path.join(root, "cap.json"), "w"))
    json.dump(grd, open(os.path.join(root, "grd.json"), "w"))
    np.savez(os.path.join(root, "proposals.npz"), dets_num=np.asarray(dets_num, dtype=np.int64), dets_labels=np.stack(dets_labels))
    with open(os.path.join(root, "data", "vg_object_vocab.txt"), "w") as f:
        f.write("\n".join(VG_CLASSES) + "\n")
    known = [w for w in WORDS if w not in ("frisbee",)] + ["puppy", "traffic", "light"]       # "frisbee", "zebra", "crossing" are OOV
    np.savez(os.path.join(root, "glove.npz"), words=np.asarray(known), vectors=synth.normal((len(known), 300), seed, "glove"))
    return SimpleNamespace(
        batch_size=2, seq_per_img=1, seq_length=seq_length, att_feat_size=feat, feature_root=fr, seg_feature_root=sr,
        num_sampled_frm=frames, num_prop_per_frm=props_per_frame, exclude_bgd_det=True, prop_thresh=0.2, t_attn_size=t_attn,
        test_mode=False, input_dic=os.path.join(root, "dic.json"), input_json=os.path.join(root, "cap.json"),
        grd_reference=os.path.join(root, "grd.json"), proposal_h5=os.path.join(root, "proposals.npz"),
        vg_vocab_file=os.path.join(root, "data", "vg_object_vocab.txt"), glove_path=os.path.join(root, "glove.npz"))
