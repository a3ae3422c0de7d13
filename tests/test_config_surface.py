"""Config / checkpoint surface parity with the reference (fixtures captured from the reference by
tools/make_opts_fixture.py and tools/make_golden.py g4)."""
import json
import os
import pickle

import pytest
import torch
import torch.nn as nn

from cvc import opts as cvc_opts
from cvc import synth
from conftest import GOLDEN

BUILD_ONLY = {"hip_graph", "dist_backend", "warm_start_mode", "results_dir", "detectron_weights_dir", "glove_path", "vg_vocab_file"}


@pytest.fixture(scope="module")
def ref_ns():
    return json.load(open(os.path.join(GOLDEN, "opts_namespaces.json")))


@pytest.fixture(scope="module")
def surface():
    return json.load(open(os.path.join(GOLDEN, "config_surface.json")))


@pytest.mark.parametrize("preset", ["baseline", "cyclical", "code_development"])
def test_presets_parse_to_reference_namespace(ref_ns, preset):
    o = cvc_opts.parse_opt(["--path_opt", "cfgs/%s.yml" % preset])
    import yaml
    from cvc.misc.utils import update_values
    with open(cvc_opts.resolve_cfg_path(o.path_opt)) as h:
        update_values(yaml.safe_load(h), vars(o))
    mine = {k: v for k, v in vars(o).items() if k not in BUILD_ONLY}
    assert mine == ref_ns[preset]


def test_cli_bool_quirk_matches_reference(ref_ns):
    """argparse type=bool: any non-empty string is truthy (reference opts.py:171,197,205,211)."""
    o = cvc_opts.parse_opt(["--train_decoder_only", "False", "--resume", "0", "--beam_size", "3", "--cuda"])
    mine = {k: v for k, v in vars(o).items() if k not in BUILD_ONLY}
    assert mine == ref_ns["cli_quirk"]
    assert o.train_decoder_only is True and o.resume is True


def test_yaml_wins_over_cli_and_paths_are_prefixed():
    o = cvc_opts.parse_opt(["--path_opt", "cfgs/cyclical.yml", "--batch_size", "7", "--rnn_size", "64", "--seed", "9"])
    o = cvc_opts.load_cfg(o)
    assert o.batch_size == 48 and o.rnn_size == 1024 and o.seed == 1          # YAML overrides the CLI
    assert o.input_json == "data/anet/cap_anet_trainval.json"
    assert o.densecap_references == ["data/anet/anet_entities_val_1.json", "data/anet/anet_entities_val_2.json"]
    assert o.train_decoder_only is False and o.caption_consistency_loss_weight == 0.5 and o.test_mode is False


def _cpu_model(d, **over):
    from helpers import make_opts
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    return DecodeAndGroundCaptionerGVDROI(make_opts(d, **over), roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))


def test_state_dict_keys_and_shapes_equal_reference(surface):
    d = synth.CONFIGS["tiny"]
    sd = _cpu_model(d).state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == surface["state_dict"]
    # shared cells: same storage under both prefixes
    assert sd["decoder_core.att_lstm.weight_ih"].data_ptr() == sd["attended_roi_decoder_core.att_lstm.weight_ih"].data_ptr()
    # synthetic checkpoints use exactly these names
    assert set(synth.hot_path_state_dict(d, 0)) == set(surface["state_dict"])


def test_finetune_param_groups_and_dead_params():
    from cvc.trainer import build_optimizer
    d = synth.CONFIGS["tiny"]
    model = _cpu_model(d)
    o = cvc_opts.parse_opt([])
    optim = build_optimizer(model, o)
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    assert len(optim.param_groups) == len(names)                                # one group per tensor
    for n, g in zip(names, optim.param_groups):
        assert g["lr"] == pytest.approx(o.learning_rate * (0.1 if "vis_embed" in n else 1.0))


@pytest.mark.parametrize("mode", ["reference", "corrected"])
def test_warm_start_routing(surface, tmp_path, mode):
    """Stage-2 warm start: suffix match, LAST checkpoint entry wins."""
    from collections import OrderedDict
    from cvc.cycle_utils import resume_decoder_roiextractor
    from cvc.model.decoder_core import TopDownDecoderCore
    from cvc.model.captioner import PrecomputedRegionFeatures
    from helpers import make_opts
    d = synth.CONFIGS["tiny"]
    full = _cpu_model(d).state_dict()
    ckpt, order = OrderedDict(), []
    for i, (k, v) in enumerate(full.items()):
        ckpt[k] = torch.full_like(v, float(i + 1))
        order.append(k)
    assert order == list(surface["state_dict"]) or sorted(order) == sorted(surface["state_dict"])
    os.makedirs(tmp_path / "baseline")
    torch.save(ckpt, tmp_path / "baseline" / "model-best.pth")
    with open(tmp_path / "baseline" / "infos_-best.pkl", "wb") as f:
        pickle.dump({"epoch": 7}, f)
    opts = make_opts(d, checkpoint_dir=str(tmp_path) + "/", id="", resume_embed=1, resume_logit=1, resume_roi_extractor=1,
                     warm_start_mode=mode)
    dec = TopDownDecoderCore(opts)
    embed = nn.Sequential(nn.Embedding(d.V, d.E), nn.ReLU(), nn.Dropout(0.5))
    logit = nn.Linear(d.R, d.V)
    roi = PrecomputedRegionFeatures(d.DET, d.G)
    resume_decoder_roiextractor(opts, "baseline", dec, embed, logit, roi)
    assert opts.start_epoch == surface["start_epoch"] == 7
    got = {}
    for name, mod in (("decoder_core", dec), ("embed", embed), ("logit", logit), ("roi_feat_extractor", roi)):
        for k, v in mod.state_dict().items():
            got[name + "." + k] = order[int(round(float(v.reshape(-1)[0]))) - 1]
    if mode == "reference":
        assert got == surface["warm_start_routing"]
        assert got["decoder_core.soft_attn.h2attn.weight"] == "attended_roi_decoder_core.soft_attn.h2attn.weight"
    else:
        assert all(k == v for k, v in got.items())


def test_decode_sequence_matches_reference_format():
    from cvc.misc.utils import decode_sequence
    itow = {str(i): "w%d" % i for i in range(10)}
    seq = torch.tensor([[5, 7, 0, 0], [0, 3, 3, 3], [1, 2, 3, 4]])
    assert decode_sequence(itow, None, None, None, None, seq, 10, None) == ["w5 w7 ", "", "w1 w2 w3 w4"]


def test_missing_roi_extractor_is_a_loud_error():
    from helpers import make_opts
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI
    with pytest.raises(ValueError, match="roi_extractor"):
        DecodeAndGroundCaptionerGVDROI(make_opts(synth.CONFIGS["tiny"]))


def test_eval_output_files_follow_the_reference_layout(tmp_path):
    """densecap / grounding JSON written by Trainer.eval (reference trainer.py:279-286, 318-329)."""
    import argparse
    import json
    from collections import defaultdict
    from cvc.trainer import write_densecap_json, write_grounding_json
    o = argparse.Namespace(results_dir=str(tmp_path / "results"), val_split="validation", id="x1")
    pred = defaultdict(list)
    from cvc.trainer import densecap_entry
    # reference trainer.py:253-261: {'sentence', 'timestamp': [round(t, 2) ...]} looked up in opts.grd_reference's
    # ['annotations'][video]['segments'][segment]['timestamps'] -- the key the ANETcaptions evaluator reads
    stamps = {"v_abc": {"segments": {"3": {"timestamps": [12.3456, 20.0049]}}}}
    vid, entry = densecap_entry("a man rides", "v_abc_segment_03", stamps)
    assert vid == "v_abc" and entry["sentence"] == "a man rides" and entry["timestamp"] == [12.35, 20.0]
    assert set(entry) >= {"sentence", "timestamp"}
    with pytest.raises(KeyError):
        densecap_entry("x", "v_abc_segment_04", stamps)
    pred["v_abc"].append({"sentence": "a man rides", "segment": "3"})
    p = write_densecap_json(pred, o)
    assert p.endswith("results/densecap-validation-x1.json")
    d = json.load(open(p))
    assert d["version"] == "VERSION 1.0" and d["results"] == {"v_abc": [{"sentence": "a man rides", "segment": "3"}]}
    assert d["external_data"] == {"used": "true", "details": "Visual Genome for Faster R-CNN pre-training"}
    grd = {"v_abc": {"3": {"clss": ["man"], "idx_in_sent": [1], "bbox_for_all_frames": [[[0, 0, 5, 5]]]}}}
    p = write_grounding_json(grd, o)
    assert p.endswith("results/attn-gen-sent-results-validation-x1.json")
    d = json.load(open(p))
    assert d["eval_mode"] == "gen" and d["results"] == grd and d["external_data"]["used"] is True


def test_grounding_box_gather_matches_reference_logic():
    """Trainer._collect_grounding (reference trainer.py:217-248): for every generated word whose lemma is a detection
    class, the most attended proposal of each sampled frame."""
    import argparse
    from collections import defaultdict
    import torch
    from cvc.trainer import Trainer
    torch.manual_seed(0)
    B, T, nf, npf = 2, 4, 3, 5
    o = argparse.Namespace(num_sampled_frm=nf, num_prop_per_frm=npf, itow={"1": "dog", "2": "runs", "3": "ball"},
                           wtol={"dog": "dog", "runs": "run", "ball": "ball"}, wtod={"dog": 1, "ball": 2}, itod={1: "dog", 2: "ball"})
    tr = Trainer(o, None, torch.nn.Linear(1, 1), None, None, None)
    ppls = torch.rand(B, nf * npf, 7) * 100
    att = torch.rand(B, T, nf * npf)
    seq = torch.tensor([[1, 2, 3, 0], [2, 0, 1, 3]])
    b = {"ppls": ppls, "seg_id": ["v_a_segment_02", "v_b_segment_10"]}
    out = defaultdict(dict)
    tr._collect_grounding(b, seq, att, out)
    assert set(out) == {"v_a", "v_b"} and set(out["v_a"]) == {"2"} and set(out["v_b"]) == {"10"}
    ra, rb = out["v_a"]["2"], out["v_b"]["10"]
    assert ra["clss"] == ["dog", "ball"] and ra["idx_in_sent"] == [0, 2]
    assert rb["clss"] == [] and rb["idx_in_sent"] == []                    # the sentence ends at the first 0
    # reference: proposals are frame-major after the permute, i.e. proposal p of frame f sits at index p * nf + f
    per = ppls.view(B, nf, npf, 7).permute(0, 2, 1, 3)
    for k, j in enumerate(ra["idx_in_sent"]):
        ind = att[0, j].view(nf, npf).argmax(-1)
        want = torch.gather(per[0], 0, ind.view(1, nf, 1).expand(1, nf, 7))[0, :, :4]
        assert torch.allclose(torch.tensor(ra["bbox_for_all_frames"][k]), want)
