"""Command-line / YAML option surface of the reference (`opts.py`), flag for flag, so existing
launch scripts and cfg files keep working: `parse_opt()` builds the same argparse Namespace
(reference opts.py:14-162 + add_cyclical_args :165-228), `load_cfg()` applies the YAML overlay in
which the YAML wins over the CLI (reference main.py:38-42, misc/utils.py:58-63), including the
reference's `type=bool` quirk (any non-empty string is True, opts.py:171,197,205,211).

New, build-side flags are grouped at the end (`--hip_graph`, `--dist_backend`).
"""
from __future__ import annotations

import argparse
import os

import yaml

from .cycle_utils import is_code_development
from .misc.utils import update_values

CFG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "cfgs")

# (flag, kwargs) in the reference's order
_DATA = [
    ("--path_opt", dict(type=str, default="cfgs/baseline.yml")),
    ("--dataset", dict(type=str, default="anet")),
    ("--data_path", dict(type=str, default="data/anet/")),
    ("--input_json", dict(type=str, default="")),
    ("--input_dic", dict(type=str, default="")),
    ("--proposal_h5", dict(type=str, default="")),
    ("--feature_root", dict(type=str, default="")),
    ("--seg_feature_root", dict(type=str, default="")),
    ("--num_workers", dict(type=int, default=4)),
    ("--cuda", dict(action="store_true")),
    ("--mGPUs", dict(action="store_true")),
]
_MODEL = [
    ("--rnn_size", dict(type=int, default=1024)),
    ("--num_layers", dict(type=int, default=1)),            # inert: decoder depth is fixed at 2
    ("--input_encoding_size", dict(type=int, default=512)),
    ("--att_hid_size", dict(type=int, default=512)),
    ("--fc_feat_size", dict(type=int, default=3072)),
    ("--att_feat_size", dict(type=int, default=2048)),
    ("--t_attn_size", dict(type=int, default=480)),
    ("--num_sampled_frm", dict(type=int, default=10)),
    ("--num_prop_per_frm", dict(type=int, default=100)),
    ("--prop_thresh", dict(type=float, default=0.2)),
    ("--att_model", dict(type=str, default="cyclical")),
    ("--att_input_mode", dict(type=str, default="both")),
    ("--t_attn_mode", dict(type=str, default="bigru")),
    ("--enable_BUTD", dict(action="store_true")),
    ("--exclude_bgd_det", dict(action="store_true")),
    ("--w_att2", dict(type=float, default=0)),
    ("--w_cls", dict(type=float, default=0)),
]
_OPTIM = [
    ("--max_epochs", dict(type=int, default=100)),
    ("--batch_size", dict(type=int, default=10)),
    ("--grad_clip", dict(type=float, default=0.1)),
    ("--drop_prob_lm", dict(type=float, default=0.5)),
    ("--seq_per_img", dict(type=int, default=1)),
    ("--seq_length", dict(type=int, default=20)),
    ("--beam_size", dict(type=int, default=1)),
    ("--optim", dict(type=str, default="adam")),
    ("--learning_rate", dict(type=float, default=5e-4)),
    ("--learning_rate_decay_start", dict(type=int, default=1)),
    ("--learning_rate_decay_every", dict(type=int, default=3)),
    ("--learning_rate_decay_rate", dict(type=float, default=0.8)),
    ("--optim_alpha", dict(type=float, default=0.9)),
    ("--optim_beta", dict(type=float, default=0.999)),
    ("--optim_epsilon", dict(type=float, default=1e-8)),
    ("--weight_decay", dict(type=float, default=0)),
    ("--start_from", dict(type=str, default=None)),
    ("--id", dict(type=str, default="")),
]
_EVAL = [
    ("--train_split", dict(type=str, default="training")),
    ("--val_split", dict(type=str, default="validation")),
    ("--inference_only", dict(action="store_true")),
    ("--densecap_references", dict(type=str, nargs="+", default=["./data/anet/anet_entities_val_1.json",
                                                               "./data/anet/anet_entities_val_2.json"])),
    ("--densecap_verbose", dict(action="store_true")),
    ("--grd_reference", dict(type=str, default="tools/anet_entities/data/anet_entities_cleaned_class_thresh50_trainval.json")),
    ("--split_file", dict(type=str, default="tools/anet_entities/data/split_ids_anet_entities.json")),
    ("--eval_obj_grounding_gt", dict(action="store_true")),
    ("--eval_obj_grounding", dict(action="store_true")),
    ("--val_every_epoch", dict(type=int, default=1)),
    ("--checkpoint_path", dict(type=str, default="save/")),
    ("--language_eval", dict(action="store_true")),
    ("--load_best_score", dict(action="store_true")),
    ("--disp_interval", dict(type=int, default=100)),
    ("--losses_log_every", dict(type=int, default=10)),
    ("--seed", dict(type=int, default=123)),
]
_CYCLICAL = [
    ("--patience", dict(default=10, type=int)),
    ("--min_lr", dict(default=5e-6, type=float)),
    ("--train_decoder_only", dict(type=bool, default=True)),
    ("--xe_loss_weight", dict(type=float, default=0.5)),
    ("--caption_consistency_loss_weight", dict(type=float, default=0.0)),
    ("--finetune_cnn", dict(default=0, type=int)),
    ("--second_drop_prob", dict(type=float, default=0.5)),
    ("--vis_encoding_size", dict(type=int, default=2048)),
    ("--softattn_type", dict(default="additive", type=str)),
    ("--softmax_temp", dict(default=1, type=float)),
    ("--embedding_vocab_plus_1", dict(type=bool, default=False)),
    ("--global_img_in_attn_lstm", dict(default=1, type=int)),
    ("--localizer_softmax_temp", dict(type=float, default=1)),
    ("--localizer_only_groundable", dict(type=bool, default=False)),   # inert in the reference too
    ("--exp_name", dict(default="experiments_", type=str)),
    ("--resume", dict(default=False, type=bool)),
    ("--tensorboard", dict(type=int, default=1)),
    ("--tb_log_dir", dict(default="tb_logs", type=str)),
    ("--resume_decoder_exp_name", dict(default="", type=str)),
    ("--resume_embed", dict(default=0, type=int)),
    ("--resume_logit", dict(default=0, type=int)),
    ("--resume_roi_extractor", dict(default=0, type=int)),
    ("--checkpoint_dir", dict(type=str, default="save/")),
]
_BUILD = [
    ("--hip_graph", dict(type=int, default=1)),             # capture the T-step decode loop in a HIP graph
    ("--dist_backend", dict(type=str, default="rccl")),     # "rccl": own RCCL communicator for the exchange + gloo control plane; "nccl": c10d's RCCL group (eager steps); "gloo": CPU tests
    ("--warm_start_mode", dict(type=str, default="reference")),  # "reference" = suffix/last-wins, "corrected"
    ("--results_dir", dict(type=str, default="results")),   # where eval writes the densecap / grounding JSON
    ("--detectron_weights_dir", dict(type=str, default="data/detectron_weights")),  # fc7 / cls_score pickles
    ("--glove_path", dict(type=str, default="data/glove.6B.300d.npz")),     # GloVe as .npz (words, vectors): no torchtext here
    ("--vg_vocab_file", dict(type=str, default="data/vg_object_vocab.txt")),  # Visual Genome classes (reference: hard-coded)
]


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    for group in (_DATA, _MODEL, _OPTIM, _EVAL):
        for flag, kw in group:
            parser.add_argument(flag, **kw)
    add_cyclical_args(parser)
    for flag, kw in _BUILD:
        parser.add_argument(flag, **kw)
    return parser


def add_cyclical_args(parser):
    for flag, kw in _CYCLICAL:
        parser.add_argument(flag, **kw)
    return parser


def develop_args(args):
    args.path_opt = "cfgs/code_development.yml"
    return args


def parse_opt(argv=None):
    args = build_parser().parse_args(argv)
    if is_code_development():      # the reference swaps in the laptop preset on macOS (opts.py:159-160)
        args = develop_args(args)
    return args


def resolve_cfg_path(path_opt: str) -> str:
    """`cfgs/X.yml` resolves against the CWD first (reference behaviour), then the packaged presets."""
    if os.path.isfile(path_opt):
        return path_opt
    cand = os.path.join(CFG_DIR, os.path.basename(path_opt))
    if os.path.isfile(cand):
        return cand
    raise FileNotFoundError(path_opt)


def load_cfg(opt, path_opt=None):
    """YAML overlay; YAML wins over CLI values (reference main.py:38-42).  Then the data-path
    prefixing of main.py:49-55."""
    path = resolve_cfg_path(path_opt or opt.path_opt)
    with open(path, "r") as handle:
        options_yaml = yaml.safe_load(handle)
    update_values(options_yaml, vars(opt))
    for key in ("input_json", "input_dic", "seg_feature_root", "feature_root", "proposal_h5"):
        setattr(opt, key, opt.data_path + getattr(opt, key))
    opt.densecap_references = [opt.data_path + reference for reference in opt.densecap_references]
    opt.test_mode = opt.val_split in ["testing", "hidden_test"]
    return opt
