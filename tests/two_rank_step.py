"""One rank of tests/test_gpu_train.py::test_two_rank_training_step_equals_the_single_process_step_on_the_whole_batch (started twice by
torch.distributed.run; both ranks on cuda:0, exchange on the stand-in librccl named by CVC_RCCL_LIB)."""
import dataclasses
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "cyclical-visual-captioning_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np
import torch
import torch.distributed as dist

from cvc import synth
from cvc.comm import RcclComm
from cvc.distributed import GradReducer, init_from_env


def build(dev, d, seed):
    from helpers import make_opts
    from cvc import opts as cvc_opts
    from cvc.model.captioner import DecodeAndGroundCaptionerGVDROI, PrecomputedRegionFeatures
    o = cvc_opts.parse_opt([])
    for k, v in vars(make_opts(d)).items():
        setattr(o, k, v)
    o.xe_loss_weight, o.caption_consistency_loss_weight, o.learning_rate, o.batch_size = 0.5, 0.5, 2e-3, d.B
    model = DecodeAndGroundCaptionerGVDROI(o, roi_extractor=PrecomputedRegionFeatures(d.DET, d.G))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.hot_path_state_dict(d, seed).items()}, strict=False)
    return o, model.to(dev).eval()          # eval-mode dropout: both sides compute the same function


def batch_of(dev, d, seed):
    from helpers import to_dev
    f, b = to_dev(synth.clip_features(d, seed), dev), to_dev(synth.label_glue_batch(d, seed), dev)
    return (f, b["input_seq"], b["gt_seq"], b["num"].cpu(), b["proposals"], b["gt_bboxs"], b["box_mask"],
            ["v_x_segment_%02d" % i for i in range(d.B)], torch.zeros(d.B, d.N, 1), b["frm_mask"], b["sample_idx"], f["pnt_mask"][:, 1:])


def main():
    from cvc.trainer import Trainer, build_optimizer
    rank, world, _ = init_from_env("gloo")      # (the control plane; "rccl" would pick cuda:LOCAL_RANK -- this box has one GPU)
    assert world == 2
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    d = dataclasses.replace(synth.CONFIGS["tiny"], B=3)
    comm = RcclComm.from_process_group()
    assert comm.count_ranks() == 2
    # ---- this rank's shard, exchanged step
    o, model = build(dev, d, 11)
    red = GradReducer(model.named_parameters(), comm=comm)
    assert red.exchange and red.world == 2 and red.backend == "rccl"
    tr = Trainer(o, None, model, build_optimizer(model, o), None, None, grad_reducer=red)
    for step in range(2):                   # step 0 learns the arenas (all buckets leave at finalize), step 1 leaves from the hooks
        tr.model.eval()
        tr.train_step(batch_of(dev, d, 100 + 10 * step + rank))
    torch.cuda.synchronize()
    mine = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    both = [torch.zeros_like(mine).cpu() for _ in range(2)]
    dist.all_gather(both, mine.cpu())
    assert torch.equal(both[0], both[1]), "ranks ended with different parameters"
    red.remove_hooks()
    b0, b1 = batch_of(dev, d, 100), batch_of(dev, d, 101)
    assert not torch.equal(b0[0]["pool_feats"], b1[0]["pool_feats"]) and not torch.equal(b0[2], b1[2]), "the two shards must differ"
    # ---- reference: ONE process, both shards, loss = mean of the two shards' loss mixes, same optimizer
    o2, ref = build(dev, d, 11)
    tr2 = Trainer(o2, None, ref, build_optimizer(ref, o2), None, None)
    for step in range(2):
        ref.eval()
        losses = []
        for r in range(2):
            b = tr2._prepare(batch_of(dev, d, 100 + 10 * step + r), True)
            losses.append(tr2.loss_mix(tr2._call(b))[0])
        tr2._backward_and_update(0.5 * (losses[0] + losses[1]))
        tr2._weights_changed()
    torch.cuda.synchronize()
    want = torch.cat([p.detach().reshape(-1) for p in ref.parameters()])
    err = float((mine - want).norm() / want.norm())
    assert err < 2e-6, err
    dist.barrier()
    comm.destroy()
    dist.destroy_process_group()
    print("TWO-RANK-STEP-OK rank", rank, "relative parameter difference to the single-process step", err, flush=True)


if __name__ == "__main__":
    main()
