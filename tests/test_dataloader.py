"""On-disk data path: cvc.misc.dataloader_anet.ANetEntitiesDataset against the REFERENCE loader's
own outputs (tests/golden/g6_dataloader.npz: misc/dataloader_anet.py::DataLoader run by tools/make_golden.py over the tiny
dataset cvc.data_fixture writes in the reference's file formats).  Every member of every 12-tuple must be identical -- dtype,
shape and bits (integer / bool work, and float values that are copies or single roundings) -- as must the constructor's GloVe
tables (including the order in which out-of-vocabulary words draw from numpy's RNG); then collation, the trainer's batch-max
trim (reference trainer.py:63-69) and, on the GPU, the pinned-memory prefetcher."""
import os
import random

import numpy as np
import pytest
import torch

from conftest import GOLDEN


@pytest.fixture(scope="module")
def g6():
    return np.load(os.path.join(GOLDEN, "g6_dataloader.npz"))


@pytest.fixture(scope="module")
def dataset_opts(tmp_path_factory):
    from cvc.data_fixture import write_tiny_anet_dataset
    return write_tiny_anet_dataset(str(tmp_path_factory.mktemp("anet")), seed=7)


def _make(o, split, test_mode):
    from cvc.misc.dataloader_anet import ANetEntitiesDataset
    o.test_mode = test_mode
    np.random.seed(3); random.seed(3)
    return ANetEntitiesDataset(o, split=split, seq_per_img=o.seq_per_img)


@pytest.mark.parametrize("tag,split,test_mode", [("train", "training", False), ("val", "validation", False), ("test", "training", True)])
def test_items_identical_to_the_reference_loader(g6, dataset_opts, tag, split, test_mode):
    ds = _make(dataset_opts, split, test_mode)
    assert len(ds) == int(g6[tag + ".len"]) and len(ds) > 0
    for i in range(len(ds)):
        item = ds[i]
        assert len(item) == 12
        for j, x in enumerate(item):
            want = g6["%s.%d.%d" % (tag, i, j)]
            if isinstance(x, str):
                assert x == str(want)
                continue
            got = x.numpy() if isinstance(x, torch.Tensor) else x
            assert got.dtype == want.dtype and got.shape == want.shape, (tag, i, j, got.dtype, want.dtype, got.shape, want.shape)
            assert np.array_equal(got, want), (tag, i, j)
    # the corner cases the fixture was built for actually occur
    it = ds[0]
    assert int(it[3][2]) >= 1 and bool(it[11][:int(it[3][1])].any()) and not bool(it[11][:int(it[3][1])].all())


def test_constructor_tables_identical_to_the_reference_loader(g6, dataset_opts):
    ds = _make(dataset_opts, "training", False)
    for k in ("glove_vg_cls", "glove_clss", "glove_w"):
        assert np.array_equal(getattr(ds, k), g6["tables." + k]), k
    assert ds.vocab_size == int(g6["tables.vocab_size"]) and ds.detect_size == int(g6["tables.detect_size"])
    assert ds.split_ix == g6["tables.split_ix"].tolist()
    # fields the reference's main.py copies onto opt (main.py:100-114)
    assert ds.wtoi["man"] == "3" and ds.wtoi["UNK"] == "1" and ds.itod[1] == "man" and ds.wtod["man"] == 1 and ds.vg_cls[0] == "__background__"
    # Trainer.eval reads the segment timestamps from the dataset (reference trainer.py:162, 259-260)
    assert ds.grd_reference["annotations"]["v_vid00"]["segments"]["0"]["timestamps"] == [2.0, 8.1]


def test_collate_and_batch_trim(dataset_opts):
    """default_collate layout of the 12-tuple, then reference trainer.py:63-69: proposals / masks / region features cut to the
    batch's largest proposal count, boxes to its largest box count."""
    from cvc.misc.dataloader_anet import collate
    ds = _make(dataset_opts, "training", False)
    batch = collate([ds[i] for i in range(len(ds))])
    B, P = len(ds), ds.max_proposal
    assert batch[0].dtype == torch.float64 and batch[0].shape == (B, ds.t_attn_size, 8)
    assert batch[4].shape == (B, P, 7) and batch[5].shape == (B, 100, 6) and batch[6].shape == (B, 1, 100, ds.seq_length + 1)
    assert batch[7] == ["v_vid00_segment_00", "v_vid01_segment_00", "v_vid01_segment_01", "v_vid03_segment_00"]
    num = batch[3]
    # the last segment overflows every padded array (dataloader_anet.py:351-352 clip to max_proposal / max_gt_box = 100): its
    # counts are the clipped ones
    assert int(num[-1, 1]) == P and int(num[-1, 2]) == 100
    sub = [x[:3] if torch.is_tensor(x) else x for x in batch]              # the ordinary segments: trimming has something to cut
    num = sub[3]
    n_prop, n_box = int(num[:, 1].max()), int(num[:, 2].max())
    assert n_prop <= P and 1 <= n_box < 100
    # everything beyond the batch maximum is padding: proposals zero / masked, boxes zero, box mask True
    assert float(sub[4][:, n_prop:].abs().max() if n_prop < P else 0) == 0 and bool(sub[11][:, n_prop:].all())
    assert float(sub[5][:, n_box:].abs().max()) == 0 and bool(sub[6][:, :, n_box:].all()) and bool(sub[9][:, :, n_box:].all())


def test_proposal_file_without_h5py_names_the_npz_twin(tmp_path):
    from cvc.misc.dataloader_anet import read_proposals
    (tmp_path / "p.h5").write_bytes(b"")
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match="npz"):
            read_proposals(str(tmp_path / "p.h5"))


@pytest.mark.gpu
def test_on_disk_batches_through_the_prefetcher_into_a_train_step(dataset_opts):
    """files -> ANetEntitiesDataset -> DataLoader(collate) -> DevicePrefetcher (pinned, side stream) -> Trainer._prepare trim:
    tensors arrive on the GPU with the trimmed shapes and the same values."""
    from torch.utils.data import DataLoader
    from cvc.misc.dataloader_anet import collate
    from cvc.prefetch import DevicePrefetcher
    from cvc.trainer import Trainer
    dev = torch.device("cuda:0")
    ds = _make(dataset_opts, "training", False)
    loader = DataLoader(ds, batch_size=2, shuffle=False, collate_fn=collate, drop_last=False)

    class _M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1, device=dev))
    tr = Trainer(dataset_opts, ds, _M(), None, loader, loader)
    host = list(loader)
    got = list(DevicePrefetcher(loader, lambda raw: tr._prepare(raw, True), dev))
    assert len(got) == len(host) == 2
    for b, raw in zip(got, host):
        n_prop, n_box = int(raw[3][:, 1].max()), int(raw[3][:, 2].max())
        assert b["ppls"].is_cuda and b["ppls"].shape == (raw[4].shape[0], n_prop, 7)
        assert b["gt_bboxs"].shape[1] == n_box and b["mask_bboxs"].shape[2] == n_box and b["mask_frms"].shape[1:] == (n_prop, n_box)
        assert b["segs_feat"].dtype == torch.float32 and torch.equal(b["segs_feat"].cpu(), raw[0].float())
        assert torch.equal(b["ppls"].cpu(), raw[4][:, :n_prop]) and torch.equal(b["ppls_feat"].cpu(), raw[8][:, :n_prop])
        assert b["pnt_mask"].shape == (raw[4].shape[0], n_prop + 1) and not bool(b["pnt_mask"][:, 0].any())
        assert b["seg_id"] == raw[7]


@pytest.mark.gpu
def test_main_trains_and_evaluates_from_disk(tmp_path):
    """cvc.main over the on-disk dataset: files -> loader -> raw frame / region features through the mirrored encoder -> the HIP
    hot path; one epoch of training, evaluation, checkpoints and the densecap file with the annotation file's timestamps."""
    import json
    from cvc import main as cvc_main
    from cvc import synth
    from cvc.data_fixture import write_tiny_anet_dataset, CLASSES, VG_CLASSES
    root = tmp_path / "anet"
    o = write_tiny_anet_dataset(str(root), seed=7, feat=24, rgb_dim=2048, bn_dim=1024, n_videos=6)
    d = synth.Dims(G=24, DET=len(CLASSES))
    tables = synth.detectron_tables(d, 7, n_vg=len(VG_CLASSES) + 1)
    import pickle
    wdir = root / "detectron"
    wdir.mkdir()
    for k in ("fc7_w", "fc7_b", "cls_score_w", "cls_score_b"):
        pickle.dump(tables[k], open(wdir / (k + ".pkl"), "wb"))
    argv = ["--no_cfg", "--max_epochs", "1", "--batch_size", "2", "--num_workers", "0", "--seq_per_img", "1",
            "--input_dic", o.input_dic, "--input_json", o.input_json, "--grd_reference", o.grd_reference, "--proposal_h5", o.proposal_h5,
            "--feature_root", o.feature_root, "--seg_feature_root", o.seg_feature_root, "--glove_path", o.glove_path,
            "--vg_vocab_file", o.vg_vocab_file, "--detectron_weights_dir", str(wdir), "--exclude_bgd_det",
            "--num_sampled_frm", "2", "--num_prop_per_frm", "5", "--t_attn_size", "6", "--att_feat_size", "24", "--vis_encoding_size", "24",
            "--rnn_size", "32", "--att_hid_size", "16", "--input_encoding_size", "16", "--seq_length", "8",
            "--train_split", "training", "--val_split", "validation", "--tensorboard", "0", "--disp_interval", "100",
            "--checkpoint_path", str(tmp_path) + "/", "--exp_name", "disk", "--learning_rate", "0.001", "--language_eval",
            "--results_dir", str(tmp_path / "results"), "--id", "d1"]
    assert cvc_main.main(argv) == 0
    dense = json.load(open(tmp_path / "results" / "densecap-validation-d1.json"))
    grd = json.load(open(o.grd_reference))["annotations"]
    assert dense["results"]
    for vid, segs in dense["results"].items():
        for s in segs:
            assert s["timestamp"] == [round(t, 2) for t in grd[vid]["segments"][s["segment"]]["timestamps"]]
    assert (tmp_path / "disk" / "model.pth").exists()
