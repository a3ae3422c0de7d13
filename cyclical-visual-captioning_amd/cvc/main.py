#!/usr/bin/env python3
"""Entry point with the reference's control flow (main.py:37-280): options + YAML overlay, seeds, data, model,
optimizer (one group per tensor), epoch loop train -> eval -> ReduceLROnPlateau -> checkpoints (`model.pth`,
`model-best.pth`, `infos_*.pkl`, `histories_*.pkl`).  Differences: one process per GPU (torchrun) with an RCCL
gradient all-reduce instead of nn.DataParallel; the dataset is the synthetic stand-in unless a loader is injected.

  python -m cvc.main --path_opt cfgs/cyclical.yml --synthetic_clips 256 --max_epochs 2
"""
from __future__ import annotations

import os
import pickle
import random
import sys

import numpy as np
import torch
from torch.optim.lr_scheduler import ReduceLROnPlateau
from torch.utils.data import DataLoader

from . import opts as cvc_opts
from . import synth
from .data_synth import SyntheticCaptionDataset, collate
from .distributed import GradReducer, init_from_env, shard_range, exchange_comm, destroy_exchange_comm
from .misc import utils
from .model.create_model import build_model
from .trainer import Trainer, build_optimizer


LAST_TRAINER = None


def main(argv=None):
    parser = cvc_opts.build_parser()
    parser.add_argument("--synthetic_clips", type=int, default=128)
    parser.add_argument("--no_cfg", action="store_true", help="skip the YAML overlay (pure CLI)")
    parser.add_argument("--synthetic_raw", action="store_true",
                        help="feed raw frame / region features through the once-per-clip encoder (model/backbone.py) "
                             "instead of pre-extracted features")
    opt = parser.parse_args(argv)
    if not opt.no_cfg:
        opt = cvc_opts.load_cfg(opt)
    opt.test_mode = opt.val_split in ["testing", "hidden_test"]                       # reference main.py:57
    rank, world, local_rank = init_from_env(opt.dist_backend) if "RANK" in os.environ else (0, 1, 0)
    device = torch.device("cuda", local_rank) if torch.cuda.is_available() else None
    if device is None:
        raise SystemExit("cvc.main needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(device)
    random.seed(opt.seed); np.random.seed(opt.seed); torch.manual_seed(opt.seed)     # reference main.py:63-67

    per_rank_bs = max(1, opt.batch_size // world)
    if opt.input_dic:
        # ---- the reference's on-disk dataset (misc/dataloader_anet.py; main.py:78-114): .npy features, proposal arrays,
        # annotation JSONs -> 12-tuples -> raw features through the once-per-clip encoder
        from .misc.dataloader_anet import ANetEntitiesDataset, collate as anet_collate
        full = ANetEntitiesDataset(opt, split=opt.train_split, seq_per_img=opt.seq_per_img)
        val_full = ANetEntitiesDataset(opt, split=opt.val_split, seq_per_img=opt.seq_per_img, num_proposals=full.num_proposals,
                                       label_proposals=full.label_proposals)
        sl = shard_range(len(full), rank, world, equal=True)
        vsl = shard_range(len(val_full), rank, world)
        loader = DataLoader(torch.utils.data.Subset(full, range(sl.start, sl.stop)), batch_size=per_rank_bs, shuffle=True,
                            num_workers=opt.num_workers, collate_fn=anet_collate, drop_last=True)
        val_loader = DataLoader(torch.utils.data.Subset(val_full, range(vsl.start, vsl.stop)), batch_size=per_rank_bs, shuffle=False,
                                num_workers=opt.num_workers, collate_fn=anet_collate)
        opt.vocab_size, opt.detect_size = full.vocab_size, full.detect_size                # reference main.py:100-114
        opt.glove_w, opt.glove_vg_cls, opt.glove_clss = (torch.from_numpy(getattr(full, k)).float()
                                                         for k in ("glove_w", "glove_vg_cls", "glove_clss"))
        for k in ("wtoi", "itow", "itod", "ltow", "itoc", "wtol", "wtod", "vg_cls"):
            setattr(opt, k, getattr(full, k))
    else:
        dims = synth.Dims(B=opt.batch_size, N=opt.num_prop_per_frm, F=opt.t_attn_size, R=opt.rnn_size, A=opt.att_hid_size,
                          E=opt.input_encoding_size, T=opt.seq_length, G=opt.vis_encoding_size, K=min(8, opt.num_prop_per_frm))
        full = SyntheticCaptionDataset(dims, opt.synthetic_clips, opt.seed, opt.train_split, raw=opt.synthetic_raw)
        # clips are sharded across ranks; every rank gets the SAME number of clips (the remainder is dropped) so that all
        # ranks run the same number of steps and issue the same collectives
        sl = shard_range(len(full), rank, world, equal=True)
        train_set = torch.utils.data.Subset(full, range(sl.start, sl.stop))
        loader = DataLoader(train_set, batch_size=per_rank_bs, shuffle=True, num_workers=0, collate_fn=collate, drop_last=True)
        val_loader = DataLoader(train_set, batch_size=per_rank_bs, shuffle=False, num_workers=0, collate_fn=collate)
        # fields the reference injects into opt from the dataset (main.py:100-114)
        opt.vocab_size, opt.itow, opt.wtoi, opt.itod, opt.detect_size = full.vocab_size, full.itow, full.wtoi, full.itod, dims.DET
        if opt.synthetic_raw:
            if opt.att_feat_size != opt.vis_encoding_size:
                raise SystemExit("--synthetic_raw needs att_feat_size == vis_encoding_size (fc7 is square, backbone.py:115)")
            opt.glove_clss, opt.glove_vg_cls = torch.from_numpy(full.glove_clss), torch.from_numpy(full.glove_vg_cls)
            opt.vg_cls, opt.detectron_tables = full.vg_cls, full.tables

    model = build_model(opt, device)
    save_dir = os.path.join(opt.checkpoint_path, opt.exp_name)

    # ---- resume (reference main.py:119-163): state_dict + infos (+ histories) of `model.pth` or `model-best.pth`
    infos, histories = {}, {}
    if opt.resume:
        suffix = "-best" if opt.load_best_score else ""
        model_path = os.path.join(save_dir, "model%s.pth" % suffix)
        info_path = os.path.join(save_dir, "infos_" + opt.id + suffix + ".pkl")
        with open(info_path, "rb") as f:
            infos = pickle.load(f)
        model.load_state_dict(torch.load(model_path, map_location=device))
        hist_path = os.path.join(save_dir, "histories_" + opt.id + ".pkl")
        if os.path.isfile(hist_path):
            with open(hist_path, "rb") as f:
                histories = pickle.load(f)
        if rank == 0:
            print("resumed %s (epoch %s, best CIDEr %s)" % (model_path, infos.get("epoch"), infos.get("best_val_score")))
    elif opt.inference_only:
        raise SystemExit("--inference_only needs --resume True (a checkpoint under %s): refusing to evaluate randomly "
                         "initialised weights" % save_dir)
    best = infos.get("best_val_score", None)
    if opt.resume_decoder_exp_name != "" and not opt.resume:                           # reference main.py:156-160
        start_epoch = getattr(opt, "start_epoch", 0)
    else:
        start_epoch = infos.get("epoch", 0)
    val_result_history = histories.get("val_result_history", {})
    lr_history = histories.get("lr_history", {})

    optimizer = build_optimizer(model, opt)
    # flat gradient arenas; exchanges only when world > 1 -- on the package's own RCCL communicator (--dist_backend rccl, the
    # default: torch.distributed on gloo is the control plane only), whose collectives are captured into the step's HIP graph
    comm = exchange_comm() if (world > 1 and opt.dist_backend == "rccl") else None
    reducer = GradReducer(model.named_parameters(), comm=comm)
    trainer = Trainer(opt, full, model, optimizer, loader, val_loader, grad_reducer=reducer)
    global LAST_TRAINER
    LAST_TRAINER = trainer                                     # (tests and notebooks: graph / deferred-error statistics of the run)
    scheduler = ReduceLROnPlateau(optimizer, 'max', patience=opt.patience, min_lr=opt.min_lr)
    tb = utils.set_tb_logger(opt.tb_log_dir, opt.exp_name, opt.resume) if (rank == 0 and opt.tensorboard and not opt.inference_only) else None

    for epoch in range(start_epoch, opt.max_epochs):
        if not opt.inference_only:
            trainer.train(epoch, tb)
        if epoch % max(opt.val_every_epoch, 1) != 0:        # reference main.py:221: scheduler + checkpoints only with an eval
            continue
        stats = trainer.eval(epoch, tb)
        if opt.inference_only:
            break
        score = stats.get("CIDEr")                            # None without an external scorer
        if world > 1:                                         # only rank 0 scores: every rank must step its LR schedule alike
            box = [score]
            torch.distributed.broadcast_object_list(box, src=0)
            score = box[0]
        if score is not None:                                 # no score -> no plateau evidence: leave the LR alone
            scheduler.step(score)
        if rank == 0:
            os.makedirs(save_dir, exist_ok=True)
            best_flag = score is not None and (best is None or score > best)
            if best_flag:
                best = score
            torch.save(model.state_dict(), os.path.join(save_dir, "model.pth"))
            val_result_history[epoch] = dict(stats)
            lr_history[epoch] = optimizer.param_groups[0]["lr"]
            infos = {"iter": (epoch + 1) * max(len(loader) - 1, 0), "epoch": epoch, "best_val_score": best, "vocab": full.itow,
                     "opt": {k: v for k, v in vars(opt).items() if _picklable(v)}}
            histories = {"val_result_history": val_result_history, "loss_history": {}, "lr_history": lr_history,
                         "ss_prob_history": {}}
            with open(os.path.join(save_dir, "infos_" + opt.id + ".pkl"), "wb") as f:
                pickle.dump(infos, f)
            with open(os.path.join(save_dir, "histories_" + opt.id + ".pkl"), "wb") as f:
                pickle.dump(histories, f)
            if best_flag or not os.path.isfile(os.path.join(save_dir, "model-best.pth")):
                torch.save(model.state_dict(), os.path.join(save_dir, "model-best.pth"))
                with open(os.path.join(save_dir, "infos_" + opt.id + "-best.pkl"), "wb") as f:
                    pickle.dump(infos, f)
    if comm is not None:
        trainer._graphs.clear()                               # captured steps hold work on the communicator
        destroy_exchange_comm()
    return 0


def _picklable(v):
    try:
        pickle.dumps(v)
        return True
    except Exception:
        return False


if __name__ == "__main__":
    sys.exit(main())
